#!/bin/bash
# round 3 iteration check: whole GPU suite, x3 GEMM probe, the default bench line (no CPU leg), kernel stats
TAG=${TAG:-r03b}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
rm -f gpurun_out/parity_log.jsonl gpurun_out/named_size.json
timeout 1500 python -m pytest tests -m gpu -q --tb=short --maxfail=40 ${PYTEST_ARGS:-} > gpurun_out/pytest_$TAG.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.txt
tail -n 30 gpurun_out/pytest_$TAG.txt | cut -c1-300
if [ -n "$PROBE" ]; then timeout 600 python $PROBE > gpurun_out/probe_$TAG.txt 2>&1; cat gpurun_out/probe_$TAG.txt | cut -c1-250; fi
timeout 900 python bench.py --no-cpu-baseline --no-tokenizer --steps 5 --warmup 2 ${BENCH_ARGS:-} > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"
tail -n 1 gpurun_out/bench_$TAG.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline())
    print({k: d[k] for k in ('value', 'ms_per_step', 'peak_hbm_gb', 'step_mfma_frac')}, d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('fast_mode', {}).get('value'))
except Exception as e:
    print('bench line unreadable', e)
"
if [ -z "$NOPROF" ]; then
OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-tokenizer --no-parity --steps 3 --warmup 1 ${BENCH_ARGS:-} ) > gpurun_out/prof_${TAG}_run.log 2>&1
python tools/rocpd_stats.py $OUT/bench_results.db > gpurun_out/prof_${TAG}_stats.txt 2>&1; head -n 24 gpurun_out/prof_${TAG}_stats.txt | cut -c1-150
find $OUT -name "*.db" -size +40M -delete
fi
