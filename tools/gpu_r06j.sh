#!/bin/bash
# round 6, call j: fp16-gradient backward of the Sparse3DNA block (class 's'): its test, then the default bench line with classes 'f' and 'fs' A/B
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_named_size.py -q -x -k "fp16_gradient" --tb=short 2>&1 | tail -n 25 > gpurun_out/r06j_test.txt; cat gpurun_out/r06j_test.txt
BA="--steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity"
: > gpurun_out/r06j_ab.txt
for rnd in 1 2; do for cls in f fs fsx; do
  AMDNUWA_BWD_F16=$cls timeout 600 python bench.py $BA 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('AMDNUWA_BWD_F16=$cls', round(d['ms_per_step'], 1), 'ms', round(d['value']), 'tok/s', 'peak GB', round(d['peak_hbm_gb'], 1), {k: round(v['ms_per_step'], 1) for k, v in d['roofline']['families'].items()})" >> gpurun_out/r06j_ab.txt
done; done
cat gpurun_out/r06j_ab.txt
