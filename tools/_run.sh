python -m pytest tests/test_gpu_vae.py -x -q 2>&1 | tail -5
python tools/vae_bench.py 2>&1 | tail -10
