#!/usr/bin/env python
"""The staggered 256x256 NT ring as a PERSISTENT kernel (one workgroup per CU walks the tile list, the DMA ring runs on across tile
borders: gemm_nt_256p_kernel, tuning key 20 = 2; `auto` takes it for K <= 1024) against the one-tile-per-workgroup launch (key 20 = 1): the fp16-operand forward GEMMs
of 'bf16x3-fwd' and the bf16 dgrad shapes of cfg 3 at per-GPU batch b.  us per call, A/B/A/B, outputs compared bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from attn_bench import bench  # noqa: E402

L = _lib.lib()
K.set_precision('bf16')            # plain bf16 outputs (no lo parts): what the backward GEMMs of every mode write
b = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M, D, FP = b * 2560, 512, 1376
h16 = (torch.randn(M, D, device='cuda') * 0.7).half()
wqkv = (torch.randn(1536, D, device='cuda') * 0.05).half()
w1 = (torch.randn(2 * FP, D, device='cuda') * 0.05).half()
w2 = (torch.randn(D, FP, device='cuda') * 0.05).half()
gg = (torch.randn(M, FP, device='cuda') * 0.3).half()


def bfp(*shape, s=0.1):
    return K.BF((torch.randn(*shape, device='cuda') * s).bfloat16(), None)


cases = [('qkv (bf16 + fp16 copies)      [M x 1536 x 512]', lambda: K.gemm_nt_f16ops(h16, wqkv, out_bf16=True, copy_f16=True), 2.0 * M * 1536 * D),
         ('FF1 + gate                     [M x 2752 x 512]', lambda: K.gemm_nt_f16ops(h16, w1, out_bf16=True, gate=True), 2.0 * M * 2 * FP * D),
         ('FF2 (fp32 out)                 [M x 512 x 1376]', lambda: K.gemm_nt_f16ops(gg, w2), 2.0 * M * D * FP)]
for name, N, Kd in (('dgrad to_out / q / out, bf16  ', 512, 512), ('dgrad qkv, bf16               ', 512, 1536), ('dgrad ff1, bf16               ', 512, 2752),
                    ('kv-like wide, bf16            ', 1024, 512), ('dgrad logits, bf16            ', 512, 8192)):
    a, w = bfp(M, Kd), bfp(N, Kd, s=0.05)
    cases.append((f'{name} [M x {N} x {Kd}]', (lambda a=a, w=w: K.gemm_nt(a, w, out_bf16=True)), 2.0 * M * N * Kd))
    cases.append((f'{name} [M x {N} x {Kd}] fp32 out', (lambda a=a, w=w: K.gemm_nt(a, w)), 2.0 * M * N * Kd))


def flat(out):
    res = []
    for o in (out if isinstance(out, tuple) else (out,)):
        for t in ((o.hi, getattr(o, 'lo', None), getattr(o, 'f16', None)) if hasattr(o, 'hi') else (o,)):
            if t is not None:
                res.append(t)
    return res


for name, fn, fl in cases:
    row, ref = [], None
    for v in (2, 1, 2, 1):
        L.amdnuwa_set_tuning(20, v)
        cur = [t.clone() for t in flat(fn())]
        if ref is None:
            ref = cur
        same = all(torch.equal(x, y) for x, y in zip(cur, ref))
        t = bench(fn, 10)
        row.append(f'{"one-tile" if v == 1 else "persist."} {t * 1e6:7.1f} us ({fl / t / 1e12:6.0f} TF)' + ('' if same else ' MISMATCH'))
    L.amdnuwa_set_tuning(20, 0)
    print(f'{name:52s} ' + ' | '.join(row), flush=True)
