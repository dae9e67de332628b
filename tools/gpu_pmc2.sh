#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc2; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $OUT -o pmc --output-format csv -- python $OLDPWD/tools/pmc_attn.py ) > gpurun_out/pmc2_run.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES -d $OUT -o pmcb --output-format csv -- python $OLDPWD/tools/pmc_attn.py ) >> gpurun_out/pmc2_run.log 2>&1
OUT2=$PWD/gpurun_out/prof3; rm -rf $OUT2; mkdir -p $OUT2
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT2 -o bench -- python $OLDPWD/bench.py --steps 2 --warmup 1 --batch 32 --no-cpu-baseline ) > gpurun_out/prof3_run.log 2>&1
grep '"metric"' gpurun_out/prof3_run.log | cut -c1-200
ls $OUT $OUT2
