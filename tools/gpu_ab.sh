#!/bin/bash
# A/B of tuning variants inside ONE box: AB="name:tuning name2:tuning2 ..." (tuning = AMDNUWA_TUNING string, '-' for default)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
for rep in 1 2; do
for ab in ${AB:-"default:-"}; do
  name=${ab%%:*}; tun=${ab#*:}; [ "$tun" = "-" ] && tun=""
  AMDNUWA_TUNING="$tun" timeout 300 python bench.py --steps ${STEPS:-6} --warmup 2 --batch ${BATCH:-32} --no-cpu-baseline --no-tokenizer > gpurun_out/ab_$name.log 2>&1
  tail -n 1 gpurun_out/ab_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'tok/s', round(d['value']), 'ms', round(d['ms_per_step'],2), 'gemm_nt TF/s', round(d['roofline']['achieved']), 'nt share', round(d['roofline']['share_of_step'],3))"
done
done
