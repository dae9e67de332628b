timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_named_size.py -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-tokenizer --no-parity --steps 8 --warmup 2 2>&1 | tail -1 | cut -c60-200
python bench.py --no-cpu-baseline --no-tokenizer --no-parity --steps 8 --warmup 2 2>&1 | tail -1 | cut -c60-200
