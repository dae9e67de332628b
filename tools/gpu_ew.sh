#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
cd tools && timeout 300 python ew_bench.py 32 > ../gpurun_out/ew_bench.log 2>&1; grep -v amdgpu.ids ../gpurun_out/ew_bench.log | tail -n 9
cd .. && timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "layernorm or geglu or stable or gemm" --timeout 300 -p no:cacheprovider 2>&1 | tail -n 3
