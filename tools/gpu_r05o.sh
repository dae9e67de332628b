#!/bin/bash
# round 5: AMDNUWA_FUSE_LINEAR_CE_X3 = auto -- the fused logits + cross entropy only when the fp32 logits would push the device past 82 % of its memory
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python -c "import torch; print('total_memory', torch.cuda.get_device_properties(0).total_memory)"
timeout 900 python -m pytest tests/test_gpu_modules.py -q --tb=short -k "cross_entropy or fused or g5" 2>&1 | tail -n 3
for i in 1 2; do timeout 600 python tools/full_step.py --batch 128 --optimizer 2>&1 | tail -n 1; done
timeout 600 python tools/full_step.py --batch 96 --optimizer 2>&1 | tail -n 1
AMDNUWA_FUSE_LINEAR_CE_X3=0 timeout 600 python tools/full_step.py --batch 128 --optimizer 2>&1 | tail -n 1
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench.py default:', round(d['ms_per_step'],1), 'ms/step', round(d['value']), d['unit'], 'peak', d.get('peak_hbm_gb'))"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_o -o st --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tokenizer --no-parity ) > /tmp/prof_o.log 2>&1
f=$(find /tmp/prof_o -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && (grep -c "ce_fwd_reg_kernel" "$f" | sed 's/^/unfused CE kernel rows in the bench trace: /')
