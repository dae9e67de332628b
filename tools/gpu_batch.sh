#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for b in 16 32; do
AMDNUWA_TUNING="0=2,3=1" timeout 300 python bench.py --steps 3 --warmup 1 --batch $b --no-cpu-baseline > gpurun_out/bench_b$b.log 2>&1; tail -n 1 gpurun_out/bench_b$b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b=$b', d['value'], d['ms_per_step'], d['step_mfma_frac'], d['roofline']['achieved'])"
done
python -c "import torch; print(torch.cuda.max_memory_allocated())"
