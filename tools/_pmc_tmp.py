import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib
L = _lib.lib(); K.set_precision('bf16'); L.amdnuwa_set_tuning(20, 1)
M = 128 * 2560
for N, Kd in ((512, 8192), (512, 1536), (1536, 512)):
    a = (torch.randn(M, Kd, device='cuda') * 0.1).bfloat16(); w = (torch.randn(N, Kd, device='cuda') * 0.05).bfloat16()
    for _ in range(3):
        K.gemm_nt(K.BF(a, None), K.BF(w, None), out_bf16=True)
        c = a @ w.t()
    torch.cuda.synchronize()
