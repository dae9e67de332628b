#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY -d $OUT -o pmc --output-format csv -- python $OLDPWD/tools/pmc_gemm.py ) > gpurun_out/pmc_run.log 2>&1
tail -3 gpurun_out/pmc_run.log; ls $OUT | head
