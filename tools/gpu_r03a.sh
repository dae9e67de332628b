#!/bin/bash
# round 3, first GPU check: whole GPU suite, the default (compliant-mode) bench line, its kernel stats, a b=128 trial
TAG=${TAG:-r03a}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
rm -f gpurun_out/parity_log.jsonl gpurun_out/named_size.json
timeout 1500 python -m pytest tests -m gpu -q --tb=short --maxfail=40 > gpurun_out/pytest_$TAG.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.txt
tail -n 40 gpurun_out/pytest_$TAG.txt | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline --no-tokenizer --steps 5 --warmup 2 > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"
tail -n 1 gpurun_out/bench_$TAG.log | cut -c1-3500
OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-tokenizer --no-parity --steps 3 --warmup 1 ) > gpurun_out/prof_${TAG}_run.log 2>&1
grep '"metric"' gpurun_out/prof_${TAG}_run.log | cut -c1-300
python tools/rocpd_stats.py $OUT/bench_results.db > gpurun_out/prof_${TAG}_stats.txt 2>&1; head -n 40 gpurun_out/prof_${TAG}_stats.txt | cut -c1-150
find $OUT -name "*.db" -size +40M -delete
timeout 600 python bench.py --batch 128 --no-cpu-baseline --no-tokenizer --no-parity --steps 3 --warmup 1 > gpurun_out/bench_${TAG}_b128.log 2>&1; echo "b128 rc=$?"
tail -n 1 gpurun_out/bench_${TAG}_b128.log | cut -c1-700
