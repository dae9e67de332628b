#!/usr/bin/env python
"""Start-phase de-phasing of the persistent NT ring (tuning key 14 = phase step in ~0.2-us units over 8 hashed phases): the K = 512 products of
the 'bf16x3-fwd' step at cfg-3 size, standalone, back to back (so the previous launch's tail is the only natural stagger)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from attn_bench import bench  # noqa: E402

L = _lib.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M, D, FP = b * 2560, 512, 1376
K.set_precision('bf16x3-fwd')
h16 = (torch.randn(M, D, device='cuda') * 0.7).half()
wqkv = (torch.randn(1536, D, device='cuda') * 0.05).half()
w1 = (torch.randn(2 * FP, D, device='cuda') * 0.05).half()
w2T = (torch.randn(FP, D, device='cuda') * 0.05).half()
wo = (torch.randn(D, D, device='cuda') * 0.05).half()
u = (torch.randn(M, 2 * FP, device='cuda') * 0.5).to(torch.bfloat16)
cases = [('qkv (one fp16 output)', lambda: K.gemm_nt_f16ops(h16, wqkv, out_f16=True)),
         ('FF1 + gate (u bf16, gate fp16)', lambda: K.gemm_nt_f16ops(h16, w1, out_bf16=True, gate=True, gate_bf16=False)),
         ('GEGLU backward (du fp16)', lambda: K.gemm_nt_geglu_bwd16(h16, w2T, u, FP)),
         ('N = 512 dgrad (one fp16 output)', lambda: K.gemm_nt_f16ops(h16, wo, out_f16=True))]
for name, fn in cases:
    row = []
    for sk in (0, 4, 8, 16, 32, 64, 0, 16):
        L.amdnuwa_set_tuning(14, sk)
        row.append(f'skew {sk:2d}: {bench(fn, 12) * 1e6:7.1f}')
    L.amdnuwa_set_tuning(14, 0)
    print(f'{name:34s} ' + ' | '.join(row), flush=True)
