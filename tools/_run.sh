export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python tools/attn_bench.py --batch 32 2>&1 | grep -i "xattn"
python -m pytest tests/test_gpu_kernels.py -q -k "cross_attention or xattn or X_CASES or attn" --tb=short 2>&1 | tail -3
