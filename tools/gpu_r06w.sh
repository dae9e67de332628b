#!/bin/bash
# round 6, call w: counters of the fp16 GEGLU-backward GEMM alone (full kernel / main loop off)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
python tools/geglu_bwd_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06w_geglu_bwd_probe.txt
: > gpurun_out/r06w_geglu_pmc.txt
for dbg in 0 2; do
for pass in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  n=$(echo $pass | cut -d' ' -f1)
  O=$R/gpurun_out/pmc_r06w_${dbg}_$n; rm -rf $O; mkdir -p $O
  ( cd /tmp && DBG=$dbg timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $O -o pmc --output-format csv -- python $R/tools/geglu_bwd_once.py ) > gpurun_out/pmc_r06w_${dbg}_$n.log 2>&1
  echo "dbg $dbg pass [$pass] rc=$?" >> gpurun_out/r06w_geglu_pmc.txt
  python - $O/pmc_counter_collection.csv >> gpurun_out/r06w_geglu_pmc.txt <<'PY'
import csv, sys
from collections import defaultdict
a = defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1], newline='')):
        if 'gemm_nt_256p' in r['Kernel_Name']:
            a[r['Counter_Name']].append(float(r['Counter_Value']))
except Exception as e:
    print('  (no csv:', e, ')')
for k, v in a.items():
    print(f'  {k:34s} per launch {sum(v) / len(v):16.1f}  ({len(v)} launches)')
PY
done; done
cat gpurun_out/r06w_geglu_pmc.txt
find gpurun_out/pmc_r06w_* -name "*.csv" -size +2M -delete
