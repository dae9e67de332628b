#!/bin/bash
# round 6, call q: de-phasing of the persistent NT ring (tuning key 14): standalone probe, then the step with the key set
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python tools/gemm_skew_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06q_skew_probe.txt
BA="--steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity"
: > gpurun_out/r06q_ab.txt
for rnd in 1 2; do for sk in 0 8 16 32; do
  AMDNUWA_TUNING=14=$sk timeout 600 python bench.py $BA 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('key 14 = $sk', round(d['ms_per_step'], 1), 'ms', round(d['value']), 'tok/s', {k: round(v['ms_per_step'], 1) for k, v in d['roofline']['families'].items()})" >> gpurun_out/r06q_ab.txt
done; done
cat gpurun_out/r06q_ab.txt
