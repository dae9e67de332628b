#!/bin/bash
# round 5, fifth GPU call: the hi + lo ring's new epilogue (bit-identity tests), the 'nofix' variant (mix loops unpinned, no packed fp32) in the
# step and under the b = 128 stress, the whole reference step at b = 128 with and without expandable allocator segments
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -q --tb=short -k "x3 or two_mfma or f16_second_copy or geglu or g2 or g5" 2>&1 | tail -n 6
AMDNUWA_LIBRARY=$R/nuwa_pytorch_amd/lib_nofix/libamdnuwa.so timeout 600 python tools/determinism_stress.py 128 --rep 10 --only-s3 > gpurun_out/r05e_stress_nofix.txt 2>&1; tail -n 2 gpurun_out/r05e_stress_nofix.txt
for i in 1 2; do
  for v in "" nofix; do
    lib=$R/nuwa_pytorch_amd/lib${v:+_$v}/libamdnuwa.so
    AMDNUWA_LIBRARY=$lib timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('build ${v:-shipped}:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'])"
  done
done > gpurun_out/r05e_ab_nofix.txt 2>&1; cat gpurun_out/r05e_ab_nofix.txt
( timeout 600 python tools/full_step.py --batch 128 --optimizer 2>&1 | tail -n 1
  PYTORCH_HIP_ALLOC_CONF=expandable_segments:True PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True timeout 600 python tools/full_step.py --batch 128 --optimizer 2>&1 | tail -n 1 | sed 's/^/expandable_segments: /'
  timeout 600 python tools/full_step.py --batch 96 --optimizer 2>&1 | tail -n 1 ) > gpurun_out/r05e_full_step.txt 2>&1; cat gpurun_out/r05e_full_step.txt
