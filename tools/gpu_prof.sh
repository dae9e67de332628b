#!/bin/bash
# kernel-trace profile of the default bench -> gpurun_out/prof_cur_stats.txt
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/prof_cur; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $R/bench.py --steps ${STEPS:-4} --warmup 2 --batch ${BATCH:-32} --no-cpu-baseline --no-tokenizer ) > gpurun_out/prof_cur_run.log 2>&1
grep '"metric"' gpurun_out/prof_cur_run.log | cut -c1-200
python tools/rocpd_stats.py $OUT/bench_results.db --last-frac 0.6 > gpurun_out/prof_cur_stats.txt 2>&1; head -42 gpurun_out/prof_cur_stats.txt | cut -c1-150
find $OUT -name "*.db" -size +40M -delete
