#!/usr/bin/env python
"""the fp16 GEGLU-backward GEMM at cfg-3 size, a few launches, for counter passes (env DBG = tuning key 7)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
L = _lib.lib()
M, D, FP = 128 * 2560, 512, 1376
K.set_precision('bf16x3-fwd')
torch.manual_seed(0)
dy16 = (torch.randn(M, D, device='cuda') * 0.7).half()
w2T = (torch.randn(FP, D, device='cuda') * 0.05).half()
u = (torch.randn(M, 2 * FP, device='cuda') * 0.5).to(torch.bfloat16)
L.amdnuwa_set_tuning(7, int(os.environ.get('DBG', '0')))
for _ in range(4):
    K.gemm_nt_geglu_bwd16(dy16, w2T, u, FP)
torch.cuda.synchronize()
