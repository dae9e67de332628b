#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
cd tools && timeout 300 python tn_probe.py ${BATCH:-64} > ../gpurun_out/tn_probe.log 2>&1; grep -v amdgpu.ids ../gpurun_out/tn_probe.log | tail -n 10
