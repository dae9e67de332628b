#!/bin/bash
# round 5, third GPU call: the fp16-gradient FeedForward backward after the epilogue split (EPI 4 / 5), A/B in the step, and kernel traces of
# the default build and the build without packed fp32 ops (which kernels pay for it)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_named_size.py -q --tb=short -x -k "fp16_gradient or fp16_store or with_fp16_gradients or fp16_operands or decoder_layer_vs_oracle or geglu" 2>&1 | tail -n 8
for i in 1 2; do
  for v in 0 f; do
    AMDNUWA_BWD_F16=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('AMDNUWA_BWD_F16=$v:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'], 'peak GB', round(d.get('peak_hbm_gb') or 0, 1))"
  done
done > gpurun_out/r05c_ab_bwd16.txt 2>&1; cat gpurun_out/r05c_ab_bwd16.txt
for v in "" nopk; do
  lib=$R/nuwa_pytorch_amd/lib${v:+_$v}/libamdnuwa.so
  OUT=$R/gpurun_out/prof_r05c_${v:-default}; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && AMDNUWA_LIBRARY=$lib timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity ) > gpurun_out/prof_r05c_${v:-default}_run.log 2>&1
  python tools/rocpd_stats.py $OUT/bench_results.db > gpurun_out/r05c_kernel_stats_${v:-default}.txt 2>&1; head -n 3 gpurun_out/r05c_kernel_stats_${v:-default}.txt | cut -c1-150
  find $OUT -name "*.db" -size +30M -delete
done
