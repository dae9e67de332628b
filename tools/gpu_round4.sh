#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
AMDNUWA_TUNING="0=3,6=2" timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "gemm or g8 or g5 or g3" > gpurun_out/pytest_c.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_c.log; tail -n 15 gpurun_out/pytest_c.log | cut -c1-200
AMDNUWA_TUNING="0=4" timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "gemm_nt" > gpurun_out/pytest_d.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_d.log; tail -n 3 gpurun_out/pytest_d.log | cut -c1-200
for b in 8 32; do
timeout 600 python tools/gemm_bench.py --batch $b --iters 10 > gpurun_out/gemm_bench_b$b.log 2>&1; grep -v amdgpu.ids gpurun_out/gemm_bench_b$b.log
done
