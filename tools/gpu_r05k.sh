#!/bin/bash
# round 5: dS / Pm of the cross-attention backward chunk-major (1 KiB contiguous per store instruction), read by the whole-M TN kernel in planes
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -q --tb=short -k "whole_m or cross_attention or xattn or reproducible or one_sample or gemm_tn or attention" 2>&1 | tail -n 8
for v in 0 1; do
( cd /tmp && AMDNUWA_XATTN_CM=$v timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_k$v -o st --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-tokenizer --no-parity ) > /tmp/prof_k.log 2>&1
f=$(find /tmp/prof_k$v -name "*kernel_stats.csv" | head -n 1); echo "AMDNUWA_XATTN_CM=$v"; [ -n "$f" ] && grep -i "gemm_tn_wm\|xattn3_bwd" "$f" | cut -c1-200
done
line() { timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'])"; }
for i in 1 2; do
  AMDNUWA_XATTN_CM=0 line "row-major dS / Pm  "
  line "chunk-major dS / Pm"
done
