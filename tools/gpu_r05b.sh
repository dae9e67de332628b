#!/bin/bash
# round 5, second GPU call: the fp16-gradient backward of the FeedForward block (kernel tests, block test, the whole GPU suite) and what it
# gives in the step: AMDNUWA_BWD_F16 off / on back to back, default build and the build without packed fp32 ops
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py -q --tb=short -x -k "fp16_gradient or fp16_store or with_fp16_gradients or fp16_operands" 2>&1 | tail -n 15
timeout 900 python -m pytest tests/test_gpu_named_size.py -q --tb=short -x -k "fp16_gradient_backward or decoder_layer_vs_oracle" 2>&1 | tail -n 15
for i in 1 2; do
  for v in 0 f; do
    AMDNUWA_BWD_F16=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('AMDNUWA_BWD_F16=$v:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'], 'peak GB', d.get('peak_hbm_gb'))"
  done
done > gpurun_out/r05b_ab_bwd16.txt 2>&1; cat gpurun_out/r05b_ab_bwd16.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r05b_pytest.txt 2>&1; tail -n 6 gpurun_out/r05b_pytest.txt
