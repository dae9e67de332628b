#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
# correctness first: full gpu suite with default tuning, then attention/gemm subset with the glds variants
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_a.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_a.log; tail -n 6 gpurun_out/pytest_a.log
AMDNUWA_TUNING="0=2,6=1,3=1,4=1,5=1" timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "gemm or sparse3dna or cross_attention or g8 or g5" > gpurun_out/pytest_b.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_b.log; tail -n 6 gpurun_out/pytest_b.log
timeout 300 python tools/attn_bench.py --batch 8 > gpurun_out/attn_bench.log 2>&1; cat gpurun_out/attn_bench.log | grep -v amdgpu.ids
timeout 600 python tools/gemm_bench.py --batch 8 --iters 10 > gpurun_out/gemm_bench2.log 2>&1; grep -A 8 "== TN" gpurun_out/gemm_bench2.log
AMDNUWA_TUNING="0=2" timeout 300 python bench.py --steps 5 --warmup 2 --batch 8 --no-cpu-baseline > gpurun_out/bench3.log 2>&1; tail -n 1 gpurun_out/bench3.log | cut -c1-330
AMDNUWA_TUNING="0=2,6=1" timeout 300 python bench.py --steps 5 --warmup 2 --batch 8 --no-cpu-baseline > gpurun_out/bench4.log 2>&1; tail -n 1 gpurun_out/bench4.log | cut -c1-330
