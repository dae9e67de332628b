#!/bin/bash
# evidence run for profiles/: default bench.py, the same under rocprofv3 --kernel-trace --stats, then separate PMC passes
TAG=${TAG:-r02}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -n 1 gpurun_out/bench_$TAG.log | cut -c1-1800
OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-tokenizer --no-parity ) > gpurun_out/prof_${TAG}_run.log 2>&1
grep '"metric"' gpurun_out/prof_${TAG}_run.log | cut -c1-200
python tools/rocpd_stats.py $OUT/bench_results.db > gpurun_out/prof_${TAG}_stats.txt 2>&1; tail -n 40 gpurun_out/prof_${TAG}_stats.txt | cut -c1-130
find $OUT -name "*.db" -size +40M -delete
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $pass | cut -d' ' -f1)
  O=$R/gpurun_out/pmc_${TAG}_$tag; rm -rf $O; mkdir -p $O
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $pass -d $O -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-tokenizer --no-parity ) > gpurun_out/pmc_${TAG}_$tag.log 2>&1
  echo "pmc $tag rc=$?"
done
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_*/pmc_counter_collection.csv > gpurun_out/pmc_${TAG}_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_${TAG}_WRITE_SIZE/pmc_counter_collection.csv --json gpurun_out/traffic_$TAG.json --key cfg3_b${BATCH:-128} > gpurun_out/pmc_${TAG}_hbm.txt 2>&1; tail -n 2 gpurun_out/pmc_${TAG}_hbm.txt
find gpurun_out/pmc_${TAG}_* -name "*.csv" -size +8M -delete
