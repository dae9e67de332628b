#!/bin/bash
# round 6, call z1: Sparse3DNA backward (fp16-gradient form) with the item / pack passes on the matrix pipe: S3 + fp16-gradient tests, phase probe, the step twice,
# the whole reference step (tools/full_step.py was broken by a quoting error in the first evidence run)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_named_size.py -q -x -k "sparse3dna or s3 or 3dna or cfg3 or fp16_gradient or bit_reproducible or one_sample" --tb=short 2>&1 | tail -n 15 > gpurun_out/r06z1_test.txt; cat gpurun_out/r06z1_test.txt
python tools/s3q_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06z1_s3q_probe.txt
BA="--steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity"
for rnd in 1 2; do
  timeout 600 python bench.py $BA 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['ms_per_step'], 1), 'ms', round(d['value']), 'tok/s', {k: round(v['ms_per_step'], 1) for k, v in d['roofline']['families'].items()})" | tee -a gpurun_out/r06z1_bench.txt
done
( timeout 900 python tools/full_step.py --batch 128 --optimizer 2>&1 | tail -n 1; timeout 900 python tools/full_step.py --batch 96 --optimizer 2>&1 | tail -n 1 ) > gpurun_out/r06z1_full_step.txt; cat gpurun_out/r06z1_full_step.txt
