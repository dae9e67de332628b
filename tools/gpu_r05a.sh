#!/bin/bash
# round 5, first GPU call: (1) which of the two cures of the round-4 head-mix defect is the cure -- the same stress on four builds
# (packed fp32 on / off  x  mix loops pinned / free); (2) what a library WITHOUT packed fp32 ops costs in the step; (3) the full-depth
# logits error over several models / samples (tools/parity_sweep.py)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
for v in pk_nofix nopk_nofix nopk ""; do
  lib=$R/nuwa_pytorch_amd/lib${v:+_$v}/libamdnuwa.so
  echo "=== build variant '${v:-default}'" 
  AMDNUWA_LIBRARY=$lib timeout 300 python tools/determinism_stress.py 16 --only-s3 2>&1 | grep -v "^$" 
done > gpurun_out/r05a_mix_variants.txt 2>&1
grep -E "===|TOTAL" gpurun_out/r05a_mix_variants.txt
for i in 1 2; do
  for v in "" nopk; do
    lib=$R/nuwa_pytorch_amd/lib${v:+_$v}/libamdnuwa.so
    AMDNUWA_LIBRARY=$lib timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('build ${v:-default}:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'], '| fast_mode', d.get('fast_mode',{}).get('ms_per_step'))"
  done
done > gpurun_out/r05a_ab_nopk.txt 2>&1; cat gpurun_out/r05a_ab_nopk.txt
timeout 900 python tools/parity_sweep.py > gpurun_out/r05a_parity_sweep.txt 2>&1; tail -n 24 gpurun_out/r05a_parity_sweep.txt
