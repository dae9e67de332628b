timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
python tools/attn_bench.py --batch 64 2>&1 | grep "kv grads"
