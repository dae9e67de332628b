#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/prof_gen; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o gen -- python $R/tools/gen_bench.py --batch ${BATCH:-4} --tokens 48 ) > gpurun_out/prof_gen_run.log 2>&1
tail -8 gpurun_out/prof_gen_run.log | cut -c1-200
python tools/rocpd_stats.py $OUT/gen_results.db > gpurun_out/prof_gen_stats.txt 2>&1; head -40 gpurun_out/prof_gen_stats.txt | cut -c1-150
find $OUT -name "*.db" -size +40M -delete
