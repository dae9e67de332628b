#!/usr/bin/env python
"""The three fp16-operand forward GEMMs of 'bf16x3-fwd' at cfg-3 size (per-GPU batch b): default 256x256 ring (one workgroup per CU)
against the K-step 64 form with staggered wave rows (tuning key 0 = 11), and the epilogue / main-loop split (key 7); the 256x128 two-workgroup tile (key 0 = 6) and the lock-step K-step 64 form (10) it also compared were removed in round 6."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from attn_bench import bench  # noqa: E402

L = _lib.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M, D, FP = b * 2560, 512, 1376
h16 = (torch.randn(M, D, device='cuda') * 0.7).half()
wqkv = (torch.randn(1536, D, device='cuda') * 0.05).half()
w1 = (torch.randn(2 * FP, D, device='cuda') * 0.05).half()
w2 = (torch.randn(D, FP, device='cuda') * 0.05).half()
gg = (torch.randn(M, FP, device='cuda') * 0.3).half()
cases = [('qkv (bf16 + fp16 copies)', lambda: K.gemm_nt_f16ops(h16, wqkv, out_bf16=True, copy_f16=True), 2.0 * M * 1536 * D, M * D * 2 + M * 1536 * 4),
         ('FF1 + gate (u bf16, gate fp16 + bf16)', lambda: K.gemm_nt_f16ops(h16, w1, out_bf16=True, gate=True), 2.0 * M * 2 * FP * D, M * D * 2 + M * 2 * FP * 2 + M * FP * 4),
         ('FF2 (fp32 out)', lambda: K.gemm_nt_f16ops(gg, w2), 2.0 * M * D * FP, M * FP * 2 + M * D * 4)]
for name, fn, fl, by in cases:
    row = []
    ref = None
    for var in (0, 11, 0, 11):
        L.amdnuwa_set_tuning(0, var)
        out = fn()
        cur = [t.float() for t in (((out.hi, out.f16) if hasattr(out, 'hi') else out) if isinstance(out, tuple) else (out,)) if t is not None]
        if ref is None:
            ref = cur
        same = all(torch.equal(a_, b_) for a_, b_ in zip(cur, ref))
        for dbg in (0, 1):
            L.amdnuwa_set_tuning(7, dbg)
            t = bench(fn, 10)
            row.append(f'{ {0: "K32 ring", 11: "K64 staggered"}[var]} {["full", "no-st"][dbg]} {t * 1e6:7.1f}' + ('' if same else ' MISMATCH'))
        L.amdnuwa_set_tuning(7, 0)
    L.amdnuwa_set_tuning(0, 0)
    print(f'{name:40s} ' + ' | '.join(row))
