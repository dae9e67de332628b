#!/bin/bash
# round 5, seventh GPU call: cross-attention backward without the padding columns (stores of all-padding lane groups skipped, TN GEMMs stop at the last key)
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_named_size.py -q --tb=short -k "cross_attention or xattn or g2 or g5 or g8 or g9 or g10 or reproducible or one_sample or decoder_layer or cfg5" 2>&1 | tail -n 8
for i in 1 2; do
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('this tree:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'], 'peak GB', round(d['peak_hbm_gb'],1))"
done > gpurun_out/r05g_bench.txt 2>&1; cat gpurun_out/r05g_bench.txt
timeout 300 python tools/xattn_bwd_probe.py 128 2>&1 | tail -n 3
