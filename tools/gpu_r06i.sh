#!/bin/bash
# round 6, call i: xattn6 backward with the dW_th products on the matrix pipe -- cross-attention tests, then the A/B of tools/xattn6_bench.py
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "xattn or cross or attention" --tb=short 2>&1 | tail -n 15 > gpurun_out/r06i_test.txt; cat gpurun_out/r06i_test.txt
timeout 600 python tools/xattn6_bench.py --batch 128 > gpurun_out/r06i_bench.txt 2>&1; cat gpurun_out/r06i_bench.txt
