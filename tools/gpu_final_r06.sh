#!/bin/bash
# the evidence run of round 6 on ONE box (TAG=r06z): reproducibility stress at b = 128 x 10, GPU suite, parity sweep (worst of 8), FETCH / WRITE passes of the
# default config (bench.py's roofline.traffic), two SQ passes (MFMA busy, waits / LDS), the default bench line, its kernel stats, attention tools, phase
# probes, the whole reference step, side configurations (cfg 4 batch sweep).  AFTER the call: copy gpurun_out/${TAG}_traffic.json over profiles/traffic.json,
# gpurun_out/parity_sweep.json over profiles/parity_sweep.json and the gpurun_out/${TAG}_* summaries into profiles/
TAG=${TAG:-r06z}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
rm -f gpurun_out/parity_log.jsonl gpurun_out/named_size.json
timeout 900 python tools/determinism_stress.py 128 --rep 10 > gpurun_out/${TAG}_determinism_stress.txt 2>&1; tail -n 1 gpurun_out/${TAG}_determinism_stress.txt
timeout 1800 python -m pytest tests -m gpu -q --tb=short > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.txt; tail -n 4 gpurun_out/${TAG}_pytest_gpu.txt
cp gpurun_out/named_size.json gpurun_out/${TAG}_named_size.json 2>/dev/null
timeout 900 python tools/parity_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_parity_sweep.txt; tail -n 3 gpurun_out/${TAG}_parity_sweep.txt | cut -c1-200
cp gpurun_out/parity_sweep.json profiles/parity_sweep.json 2>/dev/null      # (on the box: the bench line below reads it; copy it back by hand afterwards)
BA="--steps 1 --warmup 1 --no-cpu-baseline --no-tokenizer --no-parity"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
  n=$(echo $pass | cut -d' ' -f1)
  O=$R/gpurun_out/pmc_${TAG}_$n; rm -rf $O; mkdir -p $O
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $pass -d $O -o pmc --output-format csv -- python $R/bench.py $BA ) > gpurun_out/pmc_${TAG}_$n.log 2>&1
  echo "pmc pass [$pass] rc=$?"
done
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_*/pmc_counter_collection.csv > gpurun_out/${TAG}_bench_b128_pmc_all.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_${TAG}_WRITE_SIZE/pmc_counter_collection.csv --json profiles/traffic.json --key cfg3_b128_bf16x3-fwd > gpurun_out/${TAG}_bench_b128_pmc_hbm.txt 2>&1; tail -n 1 gpurun_out/${TAG}_bench_b128_pmc_hbm.txt; cp profiles/traffic.json gpurun_out/${TAG}_traffic.json
find gpurun_out/pmc_${TAG}_* -name "*.csv" -size +8M -delete
timeout 900 python bench.py > gpurun_out/${TAG}_bench_b128.log 2>&1; tail -n 1 gpurun_out/${TAG}_bench_b128.log > gpurun_out/${TAG}_bench_b128.json; cut -c1-500 gpurun_out/${TAG}_bench_b128.json
OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-tokenizer --no-parity ) > gpurun_out/prof_${TAG}_run.log 2>&1
python tools/rocpd_stats.py $OUT/bench_results.db > gpurun_out/${TAG}_bench_b128_kernel_stats.txt 2>&1; head -n 14 gpurun_out/${TAG}_bench_b128_kernel_stats.txt | cut -c1-150
find $OUT -name "*.db" -size +40M -delete
timeout 600 python tools/attn_bench.py --batch 128 > gpurun_out/${TAG}_attn_b128.txt 2>&1; grep dilation gpurun_out/${TAG}_attn_b128.txt | cut -c1-200
timeout 600 python tools/xattn6_bench.py > gpurun_out/${TAG}_xattn6_b128.txt 2>&1; grep -v amdgpu gpurun_out/${TAG}_xattn6_b128.txt | cut -c1-250
timeout 600 python tools/s3q_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_s3q_probe.txt
timeout 300 python tools/geglu_bwd_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_geglu_bwd_probe.txt; cat gpurun_out/${TAG}_geglu_bwd_probe.txt
( timeout 900 python tools/full_step.py --batch 128 --optimizer 2>&1 | tail -n 1; timeout 900 python tools/full_step.py --batch 96 --optimizer 2>&1 | tail -n 1 ) > gpurun_out/${TAG}_full_step.txt; cat gpurun_out/${TAG}_full_step.txt
: > gpurun_out/${TAG}_side_bench_configs.txt
for spec in "--config cfg2 --batch 512" "--config cfg4 --batch 64" "--config cfg4 --batch 128" "--config cfg4 --batch 256" "--precision bf16x3 --batch 16" "--precision bf16 --batch 128"; do
  echo "# python bench.py $spec --no-cpu-baseline --no-tokenizer --no-parity --steps 4 --warmup 2" >> gpurun_out/${TAG}_side_bench_configs.txt
  timeout 900 python bench.py $spec --no-cpu-baseline --no-tokenizer --no-parity --steps 4 --warmup 2 2>/dev/null | tail -n 1 >> gpurun_out/${TAG}_side_bench_configs.txt
done
( timeout 600 python tools/cfg5_step.py --batch 64 2>&1 | tail -n 2 ) > gpurun_out/${TAG}_side_steps.txt; cat gpurun_out/${TAG}_side_steps.txt
python -c "
import json
for l in open('gpurun_out/${TAG}_side_bench_configs.txt'):
    if l.startswith('{'):
        d = json.loads(l); print(d['config']['workload'][:40], d['precision_mode'], 'b', d['config']['per_gpu_batch'], round(d['ms_per_step'], 1), 'ms', round(d['value']), 'tok/s', 'peak GB', round(d.get('peak_hbm_gb', 0), 1))
"
