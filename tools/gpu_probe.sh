#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
cd tools && timeout 300 python tn_probe.py 32 > ../gpurun_out/tn_probe.log 2>&1; grep -v amdgpu.ids ../gpurun_out/tn_probe.log | tail -n 12
cd .. && timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm" --timeout 300 -p no:cacheprovider 2>&1 | tail -n 3
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-tokenizer 2>&1 | tail -n 1 | cut -c1-400
