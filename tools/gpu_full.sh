#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_g.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_g.log; tail -n 8 gpurun_out/pytest_g.log | cut -c1-220
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_g.log 2>&1; tail -n 1 gpurun_out/bench_g.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tok/s', round(d['value']), 'ms', round(d['ms_per_step'],2), 'step_frac', round(d['step_mfma_frac'],4), 'gemm_nt TF/s', round(d['roofline']['achieved']), 'nt share', round(d['roofline']['share_of_step'],3), 'tokenizer', d.get('vae_tokenizer'))"
