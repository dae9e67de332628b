#!/bin/bash
# round 5: whole-M kernel for the batched dK / dV products of the cross attention (A/B inside one call: AMDNUWA_TUNING=25=1 = the 128-row tiles).
# (The run kept in profiles/r05i_ab_tn_whole_m.txt also had a five-stage variant as key 25 = 2 and the via-reduction form as 25 = 3; the
#  five-stage kernel was dropped, 25 = 2 is the via-reduction form now.)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -q --tb=short -k "whole_m or cross_attention or xattn or reproducible or one_sample or gemm_tn" 2>&1 | tail -n 6
line() { timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'])"; }
for i in 1 2; do
  AMDNUWA_TUNING=25=1 line "128-row tiles          "
  AMDNUWA_TUNING=25=2 line "whole-M, via reduction "
  line "whole-M (4 stages)     "
done
for v in 0; do
( cd /tmp && AMDNUWA_TUNING=25=$v timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_i$v -o wm --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tokenizer --no-parity ) > /tmp/prof_i.log 2>&1
f=$(find /tmp/prof_i$v -name "*kernel_stats.csv" | head -n 1); echo "key 25 = $v: $f"; [ -n "$f" ] && grep -i "gemm_tn_wm\|gemm_tn_glds\|xattn3_bwd\|splitk" "$f" | cut -c1-200
done
