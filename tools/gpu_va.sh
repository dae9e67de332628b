#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_decode.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "g9 or dual or video_audio" > gpurun_out/pytest_va.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_va.log; tail -n 30 gpurun_out/pytest_va.log | cut -c1-220
