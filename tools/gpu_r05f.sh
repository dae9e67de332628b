#!/bin/bash
# round 5, sixth GPU call: packed (bf16 ds | bf16 P') workspace of the Sparse3DNA backward (bit-identity test, A/B through tuning key 24), lean key / value pack
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -q --tb=short -k "sparse3dna or s3 or cross_attention or xattn or g1 or g2 or g5 or g8 or reproducible or one_sample" 2>&1 | tail -n 8
for i in 1 2; do
  for v in "24=1" ""; do
    AMDNUWA_TUNING=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('AMDNUWA_TUNING=${v:-default (packed workspace)}:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'], 'peak GB', round(d['peak_hbm_gb'],1))"
  done
done > gpurun_out/r05f_ab_packed_ws.txt 2>&1; cat gpurun_out/r05f_ab_packed_ws.txt
timeout 600 python tools/attn_bench.py --batch 128 > gpurun_out/r05f_attn_b128.txt 2>&1; grep -i "dilation\|cross" gpurun_out/r05f_attn_b128.txt | cut -c1-260
