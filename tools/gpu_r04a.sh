#!/bin/bash
# round 4, call A: GPU suite (new: multi-row 3DNA forward tiles, fp16 range guard, cfg 4 / 5 parity in the benchmarked mode, bench self-launch,
# RCCL one-rank collectives), the attention micro-benchmark with the tile A/B, the default bench line
TAG=${TAG:-r04a}
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
rm -f gpurun_out/parity_log.jsonl gpurun_out/named_size.json
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/pytest_$TAG.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.txt; tail -n 40 gpurun_out/pytest_$TAG.txt | cut -c1-220
timeout 600 python tools/attn_bench.py --batch 128 > gpurun_out/attn_$TAG.txt 2>&1; grep "dilation" gpurun_out/attn_$TAG.txt | cut -c1-400
timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -n 1 gpurun_out/bench_$TAG.log | cut -c1-600
