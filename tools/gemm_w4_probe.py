#!/usr/bin/env python
"""The 4-wave (128x128 per wave) NT kernel (tuning key 0 = 9) against the 8-wave ring (7) and the vendor library on the decoder's
plain NT shapes: bit-equality of the results, full time and time with the epilogue stores skipped (tuning key 7 = 1)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from gemm_bench import bench  # noqa: E402

L = _lib.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = b * 2560
mk = lambda r, c: K.BF((torch.randn(r, c, device='cuda') * 0.5).to(torch.bfloat16), None)
shapes = [('qkv bf16', M, 1536, 512, True), ('to_out f32', M, 512, 512, False), ('ff2 f32', M, 512, 1376, False),
          ('dgrad qkv bf16', M, 512, 1536, True), ('dgrad ff1 bf16', M, 512, 2752, True), ('dgrad ff2 bf16', M, 1376, 512, True),
          ('dgrad logits bf16', M, 512, 8192, True), ('ragged f32', 1000, 777 * 8, 96, False)]
for name, m, nn, kk, obf in shapes:
    A, Bm = mk(m, kk), mk(nn, kk)
    row, ref = [], None
    for var in (7, 9):
        L.amdnuwa_set_tuning(0, var)
        out = K.gemm_nt(A, Bm, out_bf16=obf)
        o = (out.hi if obf else out).float()
        ref = o if ref is None else ref
        ok = bool(torch.equal(o, ref))
        for dbg in (0, 1):
            L.amdnuwa_set_tuning(7, dbg)
            t = bench(lambda: K.gemm_nt(A, Bm, out_bf16=obf), 10)
            row.append(f'v{var} {["full", "no-st"][dbg]} {t * 1e6:7.1f}' + ('' if ok else ' MISMATCH'))
        L.amdnuwa_set_tuning(7, 0)
    L.amdnuwa_set_tuning(0, 0)
    tl = bench(lambda: torch.matmul(A.hi, Bm.hi.t()), 10)
    print(f'{name:18s} [{m}x{nn}x{kk}] ' + ' | '.join(row) + f' | torch.matmul {tl * 1e6:7.1f} | ideal mfma {2.0 * m * nn * kk / 2.5e15 * 1e6:6.1f} us')
