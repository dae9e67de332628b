#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm" --timeout 300 -p no:cacheprovider 2>&1 | tail -n 3
cd tools && timeout 300 python gemm_probe.py 32 ${VARS:-4,7} > ../gpurun_out/gemm_probe.log 2>&1; grep -v amdgpu.ids ../gpurun_out/gemm_probe.log | tail -n 12
