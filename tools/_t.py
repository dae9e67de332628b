import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib
from attn_bench import bench
L = _lib.lib(); K.set_precision('bf16')
M = 128 * 2560
for N, Kd in ((256, 8192), (512, 8192), (1024, 8192), (2048, 8192), (256, 2752), (512, 2752), (1024, 2752)):
    a = (torch.randn(M, Kd, device='cuda') * 0.1).bfloat16(); w = (torch.randn(N, Kd, device='cuda') * 0.05).bfloat16()
    A, W = K.BF(a, None), K.BF(w, None)
    L.amdnuwa_set_tuning(0, 0); t0 = bench(lambda: K.gemm_nt(A, W, out_bf16=True), 8)
    L.amdnuwa_set_tuning(0, 12); t1 = bench(lambda: K.gemm_nt(A, W, out_bf16=True), 8)
    L.amdnuwa_set_tuning(0, 0)
    tv = bench(lambda: a @ w.t(), 8)
    f = 2.0 * M * N * Kd / 1e12
    print(f'N={N} K={Kd}: ring {t0*1e6:7.1f} ({f/t0:5.0f} TF) | w4k {t1*1e6:7.1f} ({f/t1:5.0f} TF) | vendor {tv*1e6:7.1f} ({f/tv:5.0f} TF)', flush=True)
