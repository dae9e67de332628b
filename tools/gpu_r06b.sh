#!/bin/bash
# round 6, call b: where do the K / V images of xattn6_fwd come from?  FETCH_SIZE / L2 hit counters of tools/xattn6_bench.py
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $pass | cut -d' ' -f1)
  O=$R/gpurun_out/pmc_r06b_$n; rm -rf $O; mkdir -p $O
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pass -d $O -o pmc --output-format csv -- python $R/tools/xattn6_bench.py --batch 128 --iters 3 ) > gpurun_out/pmc_r06b_$n.log 2>&1
  echo "pmc pass [$pass] rc=$?"
done
python tools/pmc_summary.py gpurun_out/pmc_r06b_*/pmc_counter_collection.csv > gpurun_out/r06b_pmc.txt 2>&1
cat gpurun_out/r06b_pmc.txt | cut -c1-300
find gpurun_out/pmc_r06b_* -name "*.csv" -size +8M -delete
