#!/usr/bin/env python
"""The long-K NT form (tuning key 0 = 12: four waves of 128x128, K-step 64, two 64 KiB stages with a 1.5-iteration prefetch, hand-placed
main loop) against the 8-wave ring (auto) and the vendor library (torch.matmul), plain bf16 operands; us per call, results compared bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from attn_bench import bench  # noqa: E402

L = _lib.lib()
K.set_precision('bf16')
b = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M = b * 2560
shapes = [('ragged', 1000, 600, 192), ('dgrad qkv', M, 512, 1536), ('dgrad ff1', M, 512, 2752), ('K 4096', M, 512, 4096), ('dgrad logits', M, 512, 8192),
          ('N 1536 K 512', M, 1536, 512), ('N 512 K 512', M, 512, 512), ('dgrad ff2 plain', M, 2752, 1408)]
for name, m, N, Kd in shapes:
    a = (torch.randn(m, Kd, device='cuda') * 0.1).bfloat16()
    w = (torch.randn(N, Kd, device='cuda') * 0.05).bfloat16()
    A, W = K.BF(a, None), K.BF(w, None)
    row = []
    for ob in (True, False):
        fn = lambda: K.gemm_nt(A, W, out_bf16=ob)
        L.amdnuwa_set_tuning(0, 0)
        ref = fn()
        ref = (ref.hi if ob else ref).clone()
        t0 = bench(fn, 10)
        L.amdnuwa_set_tuning(0, 12)
        out = fn()
        out = out.hi if ob else out
        same = torch.equal(out, ref)
        t1 = bench(fn, 10)
        L.amdnuwa_set_tuning(0, 0)
        row.append(f'{"bf16" if ob else "fp32"} out: ring {t0 * 1e6:7.1f} | w4k {t1 * 1e6:7.1f} ({2.0 * m * N * Kd / t1 / 1e12:5.0f} TF)' + ('' if same else ' MISMATCH'))
    tv = bench(lambda: a @ w.t(), 10)
    print(f'{name:16s} [{m} x {N} x {Kd}] ' + ' | '.join(row) + f' | vendor {tv * 1e6:7.1f}', flush=True)
