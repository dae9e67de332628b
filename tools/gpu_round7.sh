#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_vae.py -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_vae.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_vae.log; tail -n 25 gpurun_out/pytest_vae.log | cut -c1-250
timeout 300 python tools/vae_bench.py > gpurun_out/vae_bench.log 2>&1; grep -v amdgpu.ids gpurun_out/vae_bench.log | tail -n 12
