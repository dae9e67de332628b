#!/usr/bin/env python
"""fp16 GEGLU-backward GEMM (dy16 [M, 512] x W2^T -> du fp16 [M, 2752], the gate's backward in the epilogue) at cfg-3 size: the persistent 256x256
ring (default) against the 256x128 tile with two workgroups per CU (tuning key 0 = 6), each with the epilogue / main-loop split of tuning key 7
(bit 0: no epilogue stores, bit 1: no main loop)."""
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
torch.manual_seed(0)
dy16 = (torch.randn(M, D, device='cuda') * 0.7).half()
w2T = (torch.randn(FP, D, device='cuda') * 0.05).half()
u = (torch.randn(M, 2 * FP, device='cuda') * 0.5).to(torch.bfloat16)
fn = lambda: K.gemm_nt_geglu_bwd16(dy16, w2T, u, FP)
ref = None
for rnd in range(2):
    for var in (0,):
        L.amdnuwa_set_tuning(0, var)
        out = fn()
        if ref is None:
            ref = out.clone()
        same = torch.equal(out, ref)
        row = []
        for dbg, nm in ((0, 'full'), (1, 'no-st'), (2, 'no-ml'), (3, 'no-ml no-st')):
            L.amdnuwa_set_tuning(7, dbg)
            row.append(f'{nm} {bench(fn, 10) * 1e6:7.1f}')
        L.amdnuwa_set_tuning(7, 0)
        print(f'key 0 = {var}: ' + ' | '.join(row) + ('   same bits as the default' if same else '   DIFFERENT from the default'), flush=True)
L.amdnuwa_set_tuning(0, 0)
