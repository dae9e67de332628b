#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof2; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && AMDNUWA_TUNING="${TUNING:-0=2,3=1}" timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline ) > gpurun_out/prof2_run.log 2>&1
grep '"metric"' gpurun_out/prof2_run.log | cut -c1-250
