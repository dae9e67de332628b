#!/bin/bash
# round 5: does the persistent ring gain from its exact epilogue store counts?  AMDNUWA_TUNING=7=128 (a free probe bit) makes every count "unknown":
# the next tile's first K-steps then wait for ALL stores of the tile before (after r05i_ab_ns_full.txt: knowing the count made the GEGLU backward slower)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
for v in 0 128; do
( cd /tmp && AMDNUWA_TUNING=7=$v timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_j$v -o st --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-tokenizer --no-parity ) > /tmp/prof_j.log 2>&1
f=$(find /tmp/prof_j$v -name "*kernel_stats.csv" | head -n 1); echo "key 7 = $v"; [ -n "$f" ] && grep -i "gemm_nt_256p" "$f" | cut -c1-200
done
line() { timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'])"; }
for i in 1 2; do
  AMDNUWA_TUNING=7=128 line "counts unknown"
  line "exact counts  "
done
