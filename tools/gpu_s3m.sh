#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -p no:cacheprovider -k "sparse3dna" > gpurun_out/pytest_s3m.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_s3m.log; tail -n 25 gpurun_out/pytest_s3m.log | cut -c1-220
timeout 300 python tools/attn_bench.py --batch 16 > gpurun_out/attn_s3m.log 2>&1; head -5 gpurun_out/attn_s3m.log
