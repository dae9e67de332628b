#!/bin/bash
# after gpu_final_r03.sh and the merge of its traffic entry: the default bench line once more (now with `roofline.traffic`), and the
# FETCH / WRITE passes of the side entries (fast bf16 mode of cfg 3, cfg 2) on the same kernel sources
TAG=${TAG:-r03z}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 900 python bench.py > gpurun_out/bench_${TAG}b.log 2>&1; tail -n 1 gpurun_out/bench_${TAG}b.log | cut -c1-300
for spec in "cfg3_bf16:--precision bf16:cfg3_b128_bf16" "cfg2:--config cfg2:cfg2_b256_bf16x3-fwd"; do
  name=${spec%%:*}; rest=${spec#*:}; args=${rest%%:*}; key=${rest#*:}
  for pass in FETCH_SIZE WRITE_SIZE; do
    O=$R/gpurun_out/pmc_${TAG}_${name}_$pass; rm -rf $O; mkdir -p $O
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pass -d $O -o pmc --output-format csv -- python $R/bench.py $args --steps 1 --warmup 1 --no-cpu-baseline --no-tokenizer --no-parity ) > gpurun_out/pmc_${TAG}_${name}_$pass.log 2>&1
  done
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_${name}_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_${TAG}_${name}_WRITE_SIZE/pmc_counter_collection.csv --json gpurun_out/traffic_${TAG}_${name}.json --key $key > gpurun_out/pmc_${TAG}_${name}_hbm.txt 2>&1; tail -n 1 gpurun_out/pmc_${TAG}_${name}_hbm.txt
done
find gpurun_out/pmc_${TAG}_* -name "*.csv" -size +8M -delete
