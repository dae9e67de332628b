#!/usr/bin/env python
"""HBM-bound row kernels at the cfg-3 size (b x 2560 rows, D = 512, FP = 1376): time and achieved GB/s against the ALGORITHMIC bytes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K  # noqa: E402
from gemm_bench import bench  # noqa: E402
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
R, D, FP, n = b * 2560, 512, 1376, 2560
dev = 'cuda'
x, y, g = (torch.randn(R, D, device=dev) for _ in range(3))
w, bb = torch.randn(D, device=dev), torch.randn(D, device=dev)
def rep(name, t, by): print(f'{name:34s} {t * 1e6:8.1f} us  {by / t / 1e9:7.0f} GB/s')
t = bench(lambda: K.ln_fwd(x, w, bb), 20); rep('ln_fwd pre  (f32 -> bf16)', t, R * D * 6)
t = bench(lambda: K.ln_fwd(y, w, bb, resid=x), 20); rep('ln_fwd post (+resid, f32 -> f32)', t, R * D * 12)
h, m, r, _ = K.ln_fwd(x, w, bb)
t = bench(lambda: K.ln_bwd(g, y, m, r, w, to_bf=True, want_dsum=True), 20); rep('ln_bwd post (f32,f32 -> bf16)', t, R * D * 10)
t = bench(lambda: K.ln_bwd(g, x, m, r, w, dres=y, shift=(n, 16)), 20); rep('ln_bwd pre  (shift, +dres -> f32)', t, R * D * 16)
t = bench(lambda: K.ln_bwd(g, x, m, r, w, dres=y), 20); rep('ln_bwd pre  (+dres -> f32)', t, R * D * 16)
u = K.BF(torch.randn(R, 2 * FP, device=dev).to(torch.bfloat16), None)
dg = K.BF(torch.randn(R, FP, device=dev).to(torch.bfloat16), None)
t = bench(lambda: K.geglu_fwd(u, FP), 20); rep('geglu_fwd', t, R * FP * 6)
t = bench(lambda: K.geglu_bwd(u, dg, FP), 20); rep('geglu_bwd', t, R * FP * 10)
# chained LayerNorm backward (pre-norm of block k+1 + post-norm of block k): non-temporal variants (tuning 11) and blocks per CU (12)
from nuwa_pytorch_amd import _lib  # noqa: E402
L = _lib.lib()
dhb = K.BF(torch.randn(R, D, device=dev).to(torch.bfloat16), None)
ypb = K.BF(torch.randn(R, D, device=dev).to(torch.bfloat16), None)
m2, r2 = torch.randn(R, device=dev), torch.rand(R, device=dev) + 0.5
ref = None
for nt in (0, 1, 2, 3):
    for cap in (0, 2, 1):
        L.amdnuwa_set_tuning(11, nt); L.amdnuwa_set_tuning(12, cap)
        f = lambda: K.ln_bwd_chain(dhb, x, m, r, w, g, ypb, m2, r2, bb, shift=(n, 16), want_dsum=True)
        out = f()
        if ref is None:
            ref = out
        same = all(torch.equal(a.hi if isinstance(a, K.BF) else a, b_.hi if isinstance(b_, K.BF) else b_) for a, b_ in zip(out, ref))
        t = bench(f, 20); rep(f'ln_bwd_chain nt={nt} blocks/CU cap={cap}{"" if same else " MISMATCH"}', t, R * D * 18)
L.amdnuwa_set_tuning(11, 0); L.amdnuwa_set_tuning(12, 0)
