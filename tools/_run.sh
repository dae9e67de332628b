timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -x -q 2>&1 | tail -3
python tools/tn_probe.py 64 2>&1 | tail -8 | cut -c1-75
python bench.py --no-cpu-baseline --no-tokenizer --no-parity --steps 8 --warmup 2 2>&1 | tail -1 | cut -c60-200
