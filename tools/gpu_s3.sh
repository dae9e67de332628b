#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -q -k "sparse3dna or g1 or g8 or g5" --timeout 300 -p no:cacheprovider 2>&1 | tail -n 3
timeout 300 python tools/attn_bench.py --batch ${BATCH:-32} 2>&1 | grep -v amdgpu.ids | head -n 5
