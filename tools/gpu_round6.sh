#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_f.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_f.log; tail -n 12 gpurun_out/pytest_f.log | cut -c1-220
timeout 300 python tools/attn_bench.py --batch 8 > gpurun_out/attn_bench3.log 2>&1; grep -v amdgpu.ids gpurun_out/attn_bench3.log | head -5
for b in 8 32; do
timeout 300 python bench.py --steps 4 --warmup 2 --batch $b --no-cpu-baseline > gpurun_out/bench6_b$b.log 2>&1; tail -n 1 gpurun_out/bench6_b$b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b=$b tok/s', round(d['value']), 'ms', round(d['ms_per_step'],1), 'step_frac', round(d['step_mfma_frac'],4), 'gemm_nt TF/s', round(d['roofline']['achieved']))"
done
