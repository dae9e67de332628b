#!/bin/bash
# round 6, call z2: Sparse3DNA backward, query side: sweeps with the address arithmetic out of the plane loop (mfma_band_scores_fast / mfma_band_apply_fast)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
TAG=${TAG:-r06z2}
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_named_size.py tests/test_gpu_decode.py -q -x -k "sparse3dna or s3 or 3dna or cfg3 or fp16_gradient or bit_reproducible or one_sample or 2dna" --tb=short 2>&1 | tail -n 15 > gpurun_out/${TAG}_test.txt; cat gpurun_out/${TAG}_test.txt
python tools/s3q_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_s3q_probe.txt
timeout 600 python tools/attn_bench.py --batch 128 2>&1 | grep dilation | cut -c1-400 | tee gpurun_out/${TAG}_attn_b128.txt
BA="--steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity"
for rnd in 1 2; do
  timeout 600 python bench.py $BA 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['ms_per_step'], 1), 'ms', round(d['value']), 'tok/s', {k: round(v['ms_per_step'], 1) for k, v in d['roofline']['families'].items()})" | tee -a gpurun_out/${TAG}_bench.txt
done
