#!/bin/bash
# round 6, call zm: the whole library under the backend's other scheduling strategies (python -m nuwa_pytorch_amd.build --variant ilp / mclause:
# -mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause), A/B against the shipped build in one call
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
TAG=${TAG:-r06zm}
BA="--steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity"
for lib in default ilp mclause default ilp mclause; do
  if [ $lib = default ]; then unset AMDNUWA_LIBRARY; else export AMDNUWA_LIBRARY=$PWD/nuwa_pytorch_amd/lib_$lib/libamdnuwa.so; fi
  timeout 600 python bench.py $BA 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'], 1), 'ms', round(d['value']), 'tok/s', {k: round(v['ms_per_step'], 1) for k, v in d['roofline']['families'].items()})" | tee -a gpurun_out/${TAG}_bench.txt
done
