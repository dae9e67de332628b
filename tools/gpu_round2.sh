#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python tools/gemm_bench.py --batch 8 --iters 20 > gpurun_out/gemm_bench.log 2>&1; echo "rc=$?" >> gpurun_out/gemm_bench.log
cat gpurun_out/gemm_bench.log | tail -25
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x -k "${PYTEST_K:-layernorm or sparse3dna_core or cross_attention_core or embed or g8 or g5}" > gpurun_out/pytest_gpu3.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu3.log; tail -n 4 gpurun_out/pytest_gpu3.log
timeout 300 python bench.py --steps 5 --warmup 2 --batch 8 --no-cpu-baseline > gpurun_out/bench2.log 2>&1; tail -n 2 gpurun_out/bench2.log | cut -c1-600
