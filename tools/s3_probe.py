#!/usr/bin/env python
"""Sparse3DNA MFMA forward with phases switched off (tuning key 9; results are garbage): where does the time go?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from attn_bench import bench  # noqa: E402

L = _lib.lib()
b, n, heads, dh = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 2560, 8, 64
inner = heads * dh
torch.manual_seed(0)
qkv = K.BF(torch.randn(b * n, 3 * inner, device='cuda').to(torch.bfloat16), None)
wth = (torch.randn(heads, heads, device='cuda') * 0.3 + torch.eye(heads, device='cuda')).contiguous()
for dil in (1, 4):
    g = K.s3_geom(b, n, (10, 16, 16), (5, 3, 3), (dil, dil, dil), heads, dh)
    row = []
    for dbg, name in ((0, 'all'), (1, 'no scores'), (2, 'no softmax/mix'), (4, 'no PV'), (6, 'scores only'), (3, 'PV only'), (5, 'softmax only'), (7, 'nothing')):
        L.amdnuwa_set_tuning(9, dbg)
        t = bench(lambda: K.sparse3dna_fwd(g, qkv, wth), 10)
        row.append(f'{name} {t * 1e6:6.1f}')
    L.amdnuwa_set_tuning(9, 0)
    print(f'dilation {dil}, b={b}: ' + ' | '.join(row) + '  (us)')
