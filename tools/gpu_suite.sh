#!/bin/bash
# the whole GPU suite (+ the CPU-marked tests that need the built library), output tail to gpurun_out/${TAG}_pytest_gpu.txt
TAG=${TAG:-r06}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.txt; tail -n 30 gpurun_out/${TAG}_pytest_gpu.txt
