#!/bin/bash
# round 6, call p: Sparse3DNA backward, <bos> partials on 16-byte row pieces: the S3 tests, the phase probe, A/B of the step is implicit (r06n on another box: 476.7)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_named_size.py -q -x -k "sparse3dna or s3 or 3dna or cfg3 or fp16_gradient or bit_reproducible or one_sample" --tb=short 2>&1 | tail -n 15 > gpurun_out/r06p_test.txt; cat gpurun_out/r06p_test.txt
python tools/s3q_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06p_s3q_probe.txt
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['ms_per_step'], 1), 'ms', round(d['value']), 'tok/s', {k: round(v['ms_per_step'], 1) for k, v in d['roofline']['families'].items()})" | tee gpurun_out/r06p_bench.txt
