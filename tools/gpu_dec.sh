#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_decode.py -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/pytest_dec.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_dec.log; tail -n 40 gpurun_out/pytest_dec.log | cut -c1-250
