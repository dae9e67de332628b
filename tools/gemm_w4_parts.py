#!/usr/bin/env python
"""What bounds the 4-wave NT main loop (tuning key 0 = 9)?  Same launch with parts switched off through tuning key 7 (results are
garbage then): bit 0 stores, bit 2 MFMAs, bit 3 LDS fragment reads, bit 4 the in-loop DMA pieces."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from gemm_bench import bench  # noqa: E402

L = _lib.lib()
K.set_precision('bf16')            # plain bf16 outputs (no lo parts): what the backward GEMMs of every mode write
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = b * 2560
mk = lambda r, c: K.BF((torch.randn(r, c, device='cuda') * 0.5).to(torch.bfloat16), None)
for name, m, nn, kk in [('dgrad qkv', M, 512, 1536), ('dgrad logits', M, 512, 8192), ('qkv', M, 1536, 512)]:
    A, Bm = mk(m, kk), mk(nn, kk)
    L.amdnuwa_set_tuning(0, 9)
    row = []
    for tag, dbg in (('all', 1), ('no mfma', 1 | 4), ('no lds reads', 1 | 8), ('no dma', 1 | 16), ('mfma only', 1 | 8 | 16), ('lds reads only', 1 | 4 | 16),
                     ('dma only', 1 | 4 | 8), ('all, buffer dma', 1 | 32), ('dma only, buffer dma', 1 | 4 | 8 | 32)):
        L.amdnuwa_set_tuning(7, dbg)
        t = bench(lambda: K.gemm_nt(A, Bm, out_bf16=True), 10)
        row.append(f'{tag} {t * 1e6:7.1f}')
    L.amdnuwa_set_tuning(7, 0); L.amdnuwa_set_tuning(0, 0)
    print(f'{name:14s} [{m}x{nn}x{kk}] stores skipped: ' + ' | '.join(row) + f' | ideal mfma {2.0 * m * nn * kk / 2.5e15 * 1e6:6.1f} us')
