#!/usr/bin/env python
"""256x128 two-workgroups-per-CU NT variant (tuning 0 = 6) with a start skew for the second workgroup wave (tuning 14)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from gemm_bench import bench  # noqa: E402
L = _lib.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = b * 2560
mk = lambda r, c: K.BF((torch.randn(r, c, device='cuda') * 0.5).to(torch.bfloat16), None)
for name, m, nn, kk, obf in [('qkv bf16', M, 1536, 512, True), ('ff1-like bf16', M, 2752, 512, True), ('dgrad ff2 bf16', M, 1376, 512, True), ('to_out bf16', M, 512, 512, True),
                             ('dgrad ff1 bf16', M, 512, 2752, True)]:
    A, Bm = mk(m, kk), mk(nn, kk)
    L.amdnuwa_set_tuning(0, 7); L.amdnuwa_set_tuning(14, 0)
    ref = K.gemm_nt(A, Bm, out_bf16=obf).hi.float()
    row = []
    for var in (7, 6):
        L.amdnuwa_set_tuning(0, var)
        for skew in (0, 2, 4, 8, 12, 16, 24):
            L.amdnuwa_set_tuning(14, skew)
            ok = torch.equal(K.gemm_nt(A, Bm, out_bf16=obf).hi.float(), ref)
            row.append(f'v{var}s{skew} {bench(lambda: K.gemm_nt(A, Bm, out_bf16=obf), 10) * 1e6:6.1f}' + ('' if ok else '!'))
    L.amdnuwa_set_tuning(0, 0); L.amdnuwa_set_tuning(14, 0)
    print(f'{name:16s} [{m}x{nn}x{kk}] ' + ' | '.join(row))
