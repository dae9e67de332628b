#!/usr/bin/env python
"""a few launches of one NT and one TN GEMM shape (for rocprofv3 --pmc runs)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib
L = _lib.lib()
dev = 'cuda'
M = 20480
mk = lambda r, c: K.BF((torch.randn(r, c, device=dev) * 0.5).to(torch.bfloat16), None)
A, B = mk(M, 512), mk(1536, 512)
for v in (0, 2):
    L.amdnuwa_set_tuning(0, v)
    for _ in range(3):
        K.gemm_nt(A, B, out_bf16=True)
dY, X = mk(M, 1536), mk(M, 512)
out = torch.empty(1536, 512, device=dev)
for v in (0, 1):
    L.amdnuwa_set_tuning(6, v)
    for _ in range(3):
        K.gemm_tn(dY, X, out)
torch.cuda.synchronize()
