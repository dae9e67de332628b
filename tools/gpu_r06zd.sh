#!/bin/bash
# round 6, call zd: Sparse3DNA sweeps with no global load behind a branch (tools/probes/r06zd_s3_branchfree.patch on top of commit 135d8b9; NOT in the tree: slower), A/B against
# the library of that commit (nuwa_pytorch_amd/lib_prev/libamdnuwa.so, built from a worktree of HEAD).  Result: profiles/r06zd_s3_branchfree_ab.txt
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
TAG=${TAG:-r06zd}
PREV=$PWD/nuwa_pytorch_amd/lib_prev/libamdnuwa.so
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_named_size.py tests/test_gpu_decode.py -q -x -k "sparse3dna or s3 or 3dna or cfg3 or fp16_gradient or bit_reproducible or one_sample or 2dna" --tb=short 2>&1 | tail -n 15 > gpurun_out/${TAG}_test.txt; cat gpurun_out/${TAG}_test.txt
for lib in prev new prev new; do
  if [ $lib = prev ]; then export AMDNUWA_LIBRARY=$PREV; else unset AMDNUWA_LIBRARY; fi
  echo "== $lib" | tee -a gpurun_out/${TAG}_attn_b128.txt
  timeout 600 python tools/attn_bench.py --batch 128 2>&1 | grep dilation | cut -c1-420 | tee -a gpurun_out/${TAG}_attn_b128.txt
done
BA="--steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity"
for lib in prev new prev new; do
  if [ $lib = prev ]; then export AMDNUWA_LIBRARY=$PREV; else unset AMDNUWA_LIBRARY; fi
  timeout 600 python bench.py $BA 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'], 1), 'ms', round(d['value']), 'tok/s', {k: round(v['ms_per_step'], 1) for k, v in d['roofline']['families'].items()})" | tee -a gpurun_out/${TAG}_bench.txt
done
