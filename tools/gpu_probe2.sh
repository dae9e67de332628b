#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
cd tools && timeout 300 python gemm_probe.py 32 ${VARS:-4,6,7} > ../gpurun_out/gemm_probe.log 2>&1; grep -v amdgpu.ids ../gpurun_out/gemm_probe.log | tail -n 12
