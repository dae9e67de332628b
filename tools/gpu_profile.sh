#!/bin/bash
# rocprofv3 kernel-trace of the bench step (stats summary copied to gpurun_out/prof_*)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "${PYTEST_K:-gemm_nt or embed}" > gpurun_out/pytest_gpu2.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu2.log; tail -n 5 gpurun_out/pytest_gpu2.log
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $OLDPWD/bench.py ${BENCH_ARGS:---steps 3 --warmup 1 --batch 8 --no-cpu-baseline} ) > gpurun_out/prof_run.log 2>&1
echo "rocprof rc=$?" >> gpurun_out/prof_run.log
tail -n 4 gpurun_out/prof_run.log
find $OUT -name "*stats*" | head; 
f=$(find $OUT -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-200
# keep only the small summaries (the raw trace can be large)
find $OUT -name "*kernel_trace.csv" -size +30M -delete
