#!/bin/bash
# round 6, call y: trimmed normal-CDF (13 instructions) and packed fp16 clamp: whole GPU suite, GEGLU-backward probe, the step twice
TAG=r06y
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
TAG=$TAG bash tools/gpu_suite.sh
python tools/geglu_bwd_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_geglu_bwd_probe.txt
python tools/gemm_skew_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-60 | tee gpurun_out/${TAG}_gemm_probe.txt
BA="--steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity"
for rnd in 1 2; do
  timeout 600 python bench.py $BA 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['ms_per_step'], 1), 'ms', round(d['value']), 'tok/s', {k: round(v['ms_per_step'], 1) for k, v in d['roofline']['families'].items()})" | tee -a gpurun_out/${TAG}_bench.txt
done
