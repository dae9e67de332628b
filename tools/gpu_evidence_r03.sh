#!/bin/bash
# round-3 evidence run for profiles/ (ONE gpurun call): GPU suite, the default bench line, its rocprofv3 kernel stats, PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ) of the default config, bench lines + PMC traffic of the side configs, PMC calibration per access
# pattern, the cpu_baseline thread sweep, the side tools.   TAG=r03m bash tools/gpu_evidence_r03.sh
TAG=${TAG:-r03}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
rm -f gpurun_out/parity_log.jsonl gpurun_out/named_size.json
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_$TAG.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.txt; tail -n 4 gpurun_out/pytest_$TAG.txt
timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -n 1 gpurun_out/bench_$TAG.log | cut -c1-600
OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-tokenizer --no-parity ) > gpurun_out/prof_${TAG}_run.log 2>&1
python tools/rocpd_stats.py $OUT/bench_results.db > gpurun_out/prof_${TAG}_stats.txt 2>&1; head -n 16 gpurun_out/prof_${TAG}_stats.txt | cut -c1-140
find $OUT -name "*.db" -size +40M -delete
pmc_passes() {   # $1 = name, rest = bench args
  name=$1; shift
  for pass in "FETCH_SIZE" "WRITE_SIZE"; do
    O=$R/gpurun_out/pmc_${TAG}_${name}_$pass; rm -rf $O; mkdir -p $O
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $pass -d $O -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-tokenizer --no-parity "$@" ) > gpurun_out/pmc_${TAG}_${name}_$pass.log 2>&1
    echo "pmc $name $pass rc=$?"
  done
}
pmc_passes cfg3
O=$R/gpurun_out/pmc_${TAG}_cfg3_SQ; rm -rf $O; mkdir -p $O
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d $O -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-tokenizer --no-parity ) > gpurun_out/pmc_${TAG}_cfg3_SQ.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_cfg3_*/pmc_counter_collection.csv > gpurun_out/pmc_${TAG}_cfg3_all.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_cfg3_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_${TAG}_cfg3_WRITE_SIZE/pmc_counter_collection.csv --json gpurun_out/traffic_$TAG.json --key cfg3_b128_bf16x3-fwd > gpurun_out/pmc_${TAG}_cfg3_hbm.txt 2>&1; tail -n 1 gpurun_out/pmc_${TAG}_cfg3_hbm.txt
# side configurations: bench line + PMC traffic
: > gpurun_out/side_bench_$TAG.txt
for spec in "cfg2|--config cfg2 --batch 256|cfg2_b256_bf16x3-fwd" "cfg4|--config cfg4 --batch 8|cfg4_b8_bf16x3-fwd" "cfg3bf16|--precision bf16|cfg3_b128_bf16"; do
  name=${spec%%|*}; rest=${spec#*|}; args=${rest%%|*}; key=${rest#*|}
  echo "# python bench.py $args --no-cpu-baseline --no-tokenizer --no-parity --steps 5 --warmup 2" >> gpurun_out/side_bench_$TAG.txt
  timeout 900 python bench.py $args --no-cpu-baseline --no-tokenizer --no-parity --steps 5 --warmup 2 2>/dev/null | tail -n 1 >> gpurun_out/side_bench_$TAG.txt
  pmc_passes $name $args
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_${name}_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_${TAG}_${name}_WRITE_SIZE/pmc_counter_collection.csv --json gpurun_out/traffic_$TAG.json --key $key > gpurun_out/pmc_${TAG}_${name}_hbm.txt 2>&1; tail -n 1 gpurun_out/pmc_${TAG}_${name}_hbm.txt
done
echo "# python bench.py --precision bf16x3 --batch 16 --no-cpu-baseline --no-tokenizer --no-parity --steps 4 --warmup 2" >> gpurun_out/side_bench_$TAG.txt
timeout 900 python bench.py --precision bf16x3 --batch 16 --no-cpu-baseline --no-tokenizer --no-parity --steps 4 --warmup 2 2>/dev/null | tail -n 1 >> gpurun_out/side_bench_$TAG.txt
find gpurun_out/pmc_${TAG}_* -name "*.csv" -size +8M -delete
bash tools/gpu_calib.sh > gpurun_out/calib_${TAG}_run.log 2>&1; cat gpurun_out/calib_${TAG}.txt
timeout 300 python tools/cpu_baseline_threads.py 16,32,64 > gpurun_out/cpu_threads_$TAG.txt 2>&1; cat gpurun_out/cpu_threads_$TAG.txt
python tools/attn_bench.py --batch 64 > gpurun_out/side_attn_$TAG.txt 2>&1
python tools/gemm_probe.py 64 7 > gpurun_out/side_gemm_probe_$TAG.txt 2>&1
python tools/gemm_x3_probe.py 64 > gpurun_out/side_gemm_x3_probe_$TAG.txt 2>&1
python tools/gen_bench.py --batch 4 > gpurun_out/side_gen_$TAG.txt 2>&1
python tools/vae_bench.py > gpurun_out/side_vae_$TAG.txt 2>&1
( python tools/cfg5_step.py --batch 32; python tools/full_step.py --batch 16 ) > gpurun_out/side_steps_$TAG.txt 2>&1
tail -n 3 gpurun_out/side_steps_$TAG.txt | cut -c1-200
