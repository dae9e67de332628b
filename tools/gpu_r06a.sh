#!/bin/bash
# round 6, call a: xattn6 forward -- two SQ counter passes of tools/xattn6_bench.py.  (The phase probes this call also ran -- loop bounds
# behind tuning key 18: full 819 / no pass 1 648 / no pass 2 519 / neither 358 us -- were taken out of the kernel again: profiles/r06a_probe.txt.)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  n=$(echo $pass | cut -d' ' -f1)
  O=$R/gpurun_out/pmc_r06a_$n; rm -rf $O; mkdir -p $O
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pass -d $O -o pmc --output-format csv -- python $R/tools/xattn6_bench.py --batch 128 --iters 3 ) > gpurun_out/pmc_r06a_$n.log 2>&1
  echo "pmc pass [$pass] rc=$?"
done
python tools/pmc_summary.py gpurun_out/pmc_r06a_*/pmc_counter_collection.csv > gpurun_out/r06a_pmc.txt 2>&1
cat gpurun_out/r06a_pmc.txt | cut -c1-400
find gpurun_out/pmc_r06a_* -name "*.csv" -size +8M -delete
