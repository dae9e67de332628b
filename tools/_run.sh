timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_named_size.py -x -q 2>&1 | tail -3
python tools/attn_bench.py --batch 64 2>&1 | grep "dilation" | cut -c1-150
python bench.py --batch 64 --no-cpu-baseline --no-tokenizer --no-parity --steps 8 --warmup 2 2>&1 | tail -1 | cut -c60-200
