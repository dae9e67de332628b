#!/bin/bash
# round 6, call x: fp16 GEGLU-backward epilogue with coalesced u loads through LDS: kernel tests, fp16-gradient tests, the step twice
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_named_size.py -q -x -k "geglu or gemm or fp16_gradient or f16 or bit_reproducible or cfg3" --tb=short 2>&1 | tail -n 8 > gpurun_out/r06x_test.txt; cat gpurun_out/r06x_test.txt
BA="--steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity"
for rnd in 1 2; do
  timeout 600 python bench.py $BA 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['ms_per_step'], 1), 'ms', round(d['value']), 'tok/s', {k: round(v['ms_per_step'], 1) for k, v in d['roofline']['families'].items()})" | tee -a gpurun_out/r06x_bench.txt
done
