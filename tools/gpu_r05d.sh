#!/bin/bash
# round 5, fourth GPU call: the new tests (worst-of-8 logits, every-family reproducibility, batch-16 vs one-sample equality), the whole
# GPU suite, the b = 128 x 10 reproducibility stress, and the shipped (no packed fp32) build against the 'pk' variant in the step
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_named_size.py -q --tb=short -k "worst_of_several or every_kernel_family or equal_their_one_sample or fp16_gradient" 2>&1 | tail -n 12
timeout 900 python tools/determinism_stress.py 128 --rep 10 > gpurun_out/r05_determinism_stress.txt 2>&1; tail -n 3 gpurun_out/r05_determinism_stress.txt
for i in 1 2; do
  for v in "" pk; do
    lib=$R/nuwa_pytorch_amd/lib${v:+_$v}/libamdnuwa.so
    AMDNUWA_LIBRARY=$lib timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('build ${v:-shipped (no packed fp32)}:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'], '| fp16 saturations', d['parity']['f16_saturations'])"
  done
done > gpurun_out/r05d_ab_pk.txt 2>&1; cat gpurun_out/r05d_ab_pk.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r05d_pytest.txt 2>&1; tail -n 6 gpurun_out/r05d_pytest.txt
