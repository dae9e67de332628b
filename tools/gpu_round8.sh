#!/bin/bash
# round-1 evidence run: default bench.py under rocprofv3 --kernel-trace --stats, then separate PMC passes
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1; tail -n 1 gpurun_out/bench_default.log | cut -c1-1500
OUT=$R/gpurun_out/prof_r01; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-tokenizer ) > gpurun_out/prof_r01_run.log 2>&1
grep '"metric"' gpurun_out/prof_r01_run.log | cut -c1-300
python tools/rocpd_stats.py $OUT/bench_results.db > gpurun_out/prof_r01_stats.txt 2>&1; head -30 gpurun_out/prof_r01_stats.txt | cut -c1-170
find $OUT -name "*.db" -size +40M -delete
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $pass | cut -d' ' -f1)
  O=$R/gpurun_out/pmc_r01_$tag; rm -rf $O; mkdir -p $O
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pass -d $O -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-tokenizer ) > gpurun_out/pmc_r01_$tag.log 2>&1
  echo "pmc $tag rc=$?"; ls -la $O | head -5
done
python tools/pmc_summary.py gpurun_out/pmc_r01_*/pmc_counter_collection.csv > gpurun_out/pmc_r01_summary.txt 2>&1; head -40 gpurun_out/pmc_r01_summary.txt | cut -c1-250
# raw per-dispatch counter CSVs are large; keep only the summaries
find gpurun_out/pmc_r01_* -name "*.csv" -size +8M -delete
