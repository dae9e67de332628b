#!/bin/bash
# final check of the round on ONE box: the batch-16 bit-reproducibility stress, GPU suite, the FETCH / WRITE passes of the default config, the default bench line, its kernel stats,
# the attention / whole-step tools and (when an untracked copy of the previous round's tree sits in _ab_prev/) an A/B of the two bench lines
TAG=${TAG:-r04z}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
rm -f gpurun_out/parity_log.jsonl gpurun_out/named_size.json
timeout 300 python tools/determinism_stress.py 16 > gpurun_out/determinism_stress_$TAG.txt 2>&1; tail -n 1 gpurun_out/determinism_stress_$TAG.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_$TAG.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.txt; tail -n 4 gpurun_out/pytest_$TAG.txt
# counter passes first: the bench line below then carries the HBM traffic measured on THESE kernel sources (bench.py refuses a stale figure)
for pass in FETCH_SIZE WRITE_SIZE; do
  O=$R/gpurun_out/pmc_${TAG}_cfg3_$pass; rm -rf $O; mkdir -p $O
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $pass -d $O -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-tokenizer --no-parity ) > gpurun_out/pmc_${TAG}_cfg3_$pass.log 2>&1
done
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_cfg3_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_${TAG}_cfg3_WRITE_SIZE/pmc_counter_collection.csv --json profiles/traffic.json --key cfg3_b128_bf16x3-fwd > gpurun_out/pmc_${TAG}_cfg3_hbm.txt 2>&1; tail -n 1 gpurun_out/pmc_${TAG}_cfg3_hbm.txt; cp profiles/traffic.json gpurun_out/traffic_$TAG.json
find gpurun_out/pmc_${TAG}_* -name "*.csv" -size +8M -delete
timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -n 1 gpurun_out/bench_$TAG.log | cut -c1-400
OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-tokenizer --no-parity ) > gpurun_out/prof_${TAG}_run.log 2>&1
python tools/rocpd_stats.py $OUT/bench_results.db > gpurun_out/prof_${TAG}_stats.txt 2>&1; head -n 16 gpurun_out/prof_${TAG}_stats.txt | cut -c1-140
find $OUT -name "*.db" -size +40M -delete
timeout 600 python tools/attn_bench.py --batch 128 > gpurun_out/attn_$TAG.txt 2>&1; grep dilation gpurun_out/attn_$TAG.txt | cut -c1-330
timeout 600 python tools/attn_probe.py --batch 128 > gpurun_out/attn_probe_$TAG.txt 2>&1
timeout 900 python tools/full_step.py --batch 96 --optimizer 2>&1 | tail -n 1 > gpurun_out/full_step_$TAG.txt; timeout 900 python tools/full_step.py --batch 112 2>&1 | tail -n 1 >> gpurun_out/full_step_$TAG.txt; cat gpurun_out/full_step_$TAG.txt
if [ -d _ab_prev ]; then
  for i in 1 2; do
    ( cd _ab_prev && timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('previous round tree:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'])" )
    timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('this tree:          ', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'])"
  done > gpurun_out/ab_prev_$TAG.txt 2>&1; cat gpurun_out/ab_prev_$TAG.txt
fi
# the two-MFMA products of the compliant mode switched off / on (same tree, same box)
for i in 1 2; do
  for v in 0 default; do
    ( [ $v = default ] && unset AMDNUWA_F16X2 || export AMDNUWA_F16X2=$v; timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('AMDNUWA_F16X2=$v:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'], '| two-MFMA classes:', repr(d['fp16_forward_parts']['two_mfma_products']))" )
  done
done > gpurun_out/ab_x2_$TAG.txt 2>&1; cat gpurun_out/ab_x2_$TAG.txt
