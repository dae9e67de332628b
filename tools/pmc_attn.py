#!/usr/bin/env python
"""a few launches of the attention cores at cfg-3 geometry (for rocprofv3 --pmc runs)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib
L = _lib.lib()
dev = 'cuda'
b, n, heads, dh, T = 8, 2560, 8, 64, 256
inner = heads * dh
qkv = K.BF(torch.randn(b * n, 3 * inner, device=dev).to(torch.bfloat16), None)
do = K.BF(torch.randn(b * n, inner, device=dev).to(torch.bfloat16), None)
wth = (torch.randn(heads, heads, device=dev) * 0.3 + torch.eye(heads, device=dev)).contiguous()
g = K.s3_geom(b, n, (10, 16, 16), (5, 3, 3), (1, 1, 1), heads, dh)
for _ in range(2):
    K.sparse3dna_fwd(g, qkv, wth)
    K.sparse3dna_bwd(g, qkv, wth, do)
gx = K.x_geom(b, n, T, heads, dh)
q = K.BF(torch.randn(b * n, inner, device=dev).to(torch.bfloat16), None)
kv = K.BF(torch.randn(b * T, 2 * inner, device=dev).to(torch.bfloat16), None)
nk = torch.randn(heads, dh, device=dev)
pk = K.xattn_pack(gx, kv, nk, nk, None)
for _ in range(2):
    o, P, Pm = K.xattn_fwd(gx, q, pk, wth)
    K.xattn_bwd(gx, do, pk, wth, P)
    o2, st = K.xattn2_fwd(gx, q, pk, wth)
    K.xattn2_bwd(gx, q, do, pk, wth, st)
torch.cuda.synchronize()
