#!/bin/bash
# the two-MFMA products (fp16 activation x fp16 hi + lo weight): kernel tests, the full-depth logits with each class switched on, and an A/B of the step
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
rm -f gpurun_out/named_size.json
timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "two_mfma or hand_over or f16" > gpurun_out/x2_kernels.txt 2>&1; tail -n 6 gpurun_out/x2_kernels.txt
timeout 500 python -m pytest tests/test_gpu_named_size.py -m gpu -q -x --tb=short -k "full_depth" > gpurun_out/x2_full_depth.txt 2>&1; tail -n 6 gpurun_out/x2_full_depth.txt
python - <<'PY'
import json
d = json.load(open('gpurun_out/named_size.json'))['cfg3.full_depth_logits']
for k, v in d.items():
    print(k, {a: (round(b, 7) if isinstance(b, float) else b) for a, b in v.items() if a.startswith('logits')})
PY
for v in 0 oql 0 oql o; do
  AMDNUWA_F16X2=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('AMDNUWA_F16X2=$v:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'], 'roofline', round(d['roofline']['frac'],4))"
done 2>&1 | tee gpurun_out/x2_ab.txt
