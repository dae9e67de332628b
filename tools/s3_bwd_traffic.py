#!/usr/bin/env python
"""Sparse3DNA backward at cfg-3 geometry (dilation 2, b from argv): the default workspace form and the recomputing key side (tuning key 4 = 4),
a few calls each -- run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` to compare their HBM traffic, or plain for times."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from gemm_bench import bench  # noqa: E402

L = _lib.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, heads, dh = 2560, 8, 64
inner = heads * dh
g = K.s3_geom(b, n, (10, 16, 16), (5, 3, 3), (2, 2, 2), heads, dh)
mk = lambda c: K.BF((torch.randn(b * n, c, device='cuda') * 0.5).to(torch.bfloat16), None)
qkv, do = mk(3 * inner), mk(inner)
wth = (torch.randn(heads, heads) * 0.3 + torch.eye(heads)).cuda()
for form, key in (('workspace', 0), ('recompute', 4)):
    L.amdnuwa_set_tuning(4, key)
    t = bench(lambda: K.sparse3dna_bwd(g, qkv, wth, do), 5)
    print(f'{form:10s} backward {t * 1e6:8.1f} us  (algorithmic {8 * b * n * inner * 2 / 1e9:.2f} GB)')
L.amdnuwa_set_tuning(4, 0)
