timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vae.py -x -q 2>&1 | tail -3
python tools/ew_bench.py 64 2>&1 | sed -n 2,10p
