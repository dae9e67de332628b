#!/bin/bash
# round 6, call zk: knob sweep of the final tree in one call -- do older A/B decisions still hold after the round's kernel changes?
#   22=2  every K % 64 == 0 NT product on the four-wave K-step 64 kernel (also the K = 512 ones, which the persistent ring keeps by default)
#   20=1  no persistent ring          AMDNUWA_FUSE_LINEAR_CE_X3=1  fused to_logits + cross entropy          3=1 / 12..: see include/amdnuwa.h
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
TAG=${TAG:-r06zk}
BA="--steps 6 --warmup 2 --no-cpu-baseline --no-tokenizer --no-parity"
run() {
  timeout 600 python bench.py $BA 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'], 1), 'ms', round(d['value']), 'tok/s', {k: round(v['ms_per_step'], 1) for k, v in d['roofline']['families'].items()})" | tee -a gpurun_out/${TAG}_knobs.txt
}
run "default            "
AMDNUWA_TUNING=22=2 run "22=2 (w4k, all K)  "
AMDNUWA_TUNING=20=1 run "20=1 (no persist.) "
AMDNUWA_FUSE_LINEAR_CE_X3=1 run "fused logits + CE  "
AMDNUWA_CHAIN_BWD=0 run "no chained LN bwd  "
run "default            "
