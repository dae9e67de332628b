#!/bin/bash
# round 5: the three forms of the cross-attention dK / dV path in ONE call, kernel stats of each (boxes differ by several per cent on the
# bandwidth-bound kernels): 128-row tiles + row-major dS / P' (round 4) | whole-M kernel + row-major | whole-M kernel + chunk-major (shipped)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
run() {
( cd /tmp && env $2 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_l$1 -o st --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity ) > /tmp/prof_l.log 2>&1
f=$(find /tmp/prof_l$1 -name "*kernel_stats.csv" | head -n 1); echo "== $3  [$2]"; [ -n "$f" ] && python -c "
import csv,sys
for r in csv.reader(open('$f')):
    if any(k in r[0] for k in ('gemm_tn_wm','gemm_tn_glds','xattn3_bwd','splitk_reduce')): print('   %-40s calls %5s  total %8.1f ms  avg %8.1f us' % (r[0].split('(')[-2].split('::')[-1][:40] if '::' in r[0] else r[0][:40], r[1], float(r[2])/1e6, float(r[3])/1e3))
"
rm -rf /tmp/prof_l$1
}
for i in 1 2; do
run a$i "AMDNUWA_TUNING=25=1" "128-row tiles, row-major"
run b$i "AMDNUWA_XATTN_CM=0" "whole-M, row-major"
run c$i "AMDNUWA_XATTN_CM=1" "whole-M, chunk-major (shipped)"
done
