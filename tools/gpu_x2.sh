#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "cross_attention" --timeout 300 -p no:cacheprovider 2>&1 | grep -v "^$" | grep "Error\|assert\|rel\|xattn2\|passed\|failed" | head -n 30 | cut -c1-250
timeout 300 python tools/attn_bench.py --batch ${BATCH:-32} 2>&1 | grep -v amdgpu.ids | tail -n 3
