#!/usr/bin/env python
"""bench.py's cpu_baseline leg (the oracle's full cfg-3 decoder step, b = 1) at several host thread counts: the log behind the
32-thread cap (SURVEY.md section 8(d) asks for os.cpu_count() threads; beyond ~32 the fp32 oracle gets slower on the GPU box's host).
    python tools/cpu_baseline_threads.py [32,64,0]        (0 = os.cpu_count())"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

counts = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '32,0').split(',')]
c = bench.CFGS['cfg3']
nuwa = bench.build_model(c, 'cpu')
ids, ctx, mask = bench.synthetic_batch(c, 1, 0, 'cpu')
print(f'host cores: {os.cpu_count()}')
for n in counts:
    t = os.cpu_count() if n == 0 else n
    out, _ = bench.cpu_baseline(c, nuwa, ids, ctx, mask, max_runs=3 if t <= 64 else 1, budget_s=90.0, threads=t)
    print(f'threads {out["cores"]:4d}: {out["seconds_per_step"]:7.1f} s per step = {out["value"]:7.1f} video-tokens/s   ({out["sample"].split("median of")[-1].strip()})', flush=True)
