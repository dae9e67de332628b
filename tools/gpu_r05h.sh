#!/bin/bash
# round 5: Sparse3DNA backward without the workspace entries nobody reads (masked planes, negative key columns, the <bos> slot)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -q --tb=short -k "sparse3dna or s3 or g1 or g5 or g8 or reproducible or one_sample" 2>&1 | tail -n 6
for i in 1 2; do
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('this tree:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'])"
done
timeout 600 python tools/attn_bench.py --batch 128 2>&1 | grep "dilation" | cut -c150-330
