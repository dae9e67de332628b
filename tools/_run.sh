timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -x -q -k "s3 or sparse or 3dna or Sparse" 2>&1 | tail -3
python tools/attn_bench.py --batch 64 2>&1 | grep "dilation" | cut -c1-150
