#!/usr/bin/env python
"""TN (weight-gradient) GEMM shapes of the decoder step: staggered vs lock-step 256x256 ring."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from gemm_bench import bench  # noqa: E402
L = _lib.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M, n = b * 2560, 2560
mk = lambda r, c: K.BF((torch.randn(r, c, device='cuda') * 0.5).to(torch.bfloat16), None)
for name, n1, n2, sh in [('dW q (shift)', 512, 512, True), ('dW kv (shift)', 1024, 512, True), ('dW qkv (shift)', 1536, 512, True), ('dW out', 512, 512, False),
                         ('dW ff1 half (shift)', 1365, 512, True), ('dW ff1 full (shift)', 2752, 512, True), ('dW ff2', 512, 1365, False), ('dW logits', 8192, 512, False)]:
    ld1, ld2 = (n1 + 31) // 32 * 32, (n2 + 31) // 32 * 32
    A, Bm = mk(M, ld1), mk(M, ld2)
    out = torch.empty(n1, n2, device='cuda')
    row, ref = [], None
    for st in (1, 2):
        L.amdnuwa_set_tuning(8, st)
        f = lambda: K.gemm_tn(K.view(A, cols=slice(0, n1)), K.view(Bm, cols=slice(0, n2)), out, shift=(n, 16) if sh else None, N1=n1, N2=n2)
        f()
        if ref is None:
            ref = out.clone()
        err = float((out - ref).abs().max() / ref.abs().max())
        t = bench(f, 10)
        row.append(f'{"lockstep" if st == 1 else "stagger "} {t * 1e6:6.1f} us {2.0 * M * n1 * n2 / t / 1e12:6.1f} TF/s' + ('' if err < 1e-5 else f' MISMATCH {err:.1e}'))
    L.amdnuwa_set_tuning(8, 0)
    At, Bt = A.hi[:, :n1], Bm.hi[:, :n2]
    tl = bench(lambda: torch.matmul(At.t(), Bt), 10)          # library yardstick (no token shift)
    row.append(f'torch.matmul {tl * 1e6:6.1f} us {2.0 * M * n1 * n2 / tl / 1e12:6.1f} TF/s')
    print(f'{name:22s} [{n1}x{n2}]  ' + ' | '.join(row))
