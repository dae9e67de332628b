#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_modules.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "sketch" > gpurun_out/pytest_sk.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_sk.log; tail -n 40 gpurun_out/pytest_sk.log | cut -c1-250
