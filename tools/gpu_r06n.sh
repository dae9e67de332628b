#!/bin/bash
# round 6, call n: tree with the fp16-gradient range fix (V image 2^-6, grad scale [2^-4, 2^-3)): whole GPU suite, default bench line, kernel stats
TAG=${TAG:-r06n}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
TAG=$TAG bash tools/gpu_suite.sh
timeout 900 python bench.py --no-cpu-baseline --no-tokenizer > gpurun_out/${TAG}_bench_b128.log 2>&1; tail -n 1 gpurun_out/${TAG}_bench_b128.log > gpurun_out/${TAG}_bench_b128.json; cut -c1-300 gpurun_out/${TAG}_bench_b128.json
OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-tokenizer --no-parity ) > gpurun_out/prof_${TAG}_run.log 2>&1
python tools/rocpd_stats.py $OUT/bench_results.db > gpurun_out/${TAG}_bench_b128_kernel_stats.txt 2>&1; head -n 40 gpurun_out/${TAG}_bench_b128_kernel_stats.txt | cut -c1-160
find $OUT -name "*.db" -size +40M -delete
