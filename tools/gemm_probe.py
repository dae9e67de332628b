#!/usr/bin/env python
"""Where does the 256x256 NT kernel spend its time?  Times each decoder NT shape normally, with the epilogue stores
skipped (tuning 7 = 1), and with the main loop skipped (tuning 7 = 2)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from gemm_bench import bench  # noqa: E402

L = _lib.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
variants = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]
M, n = b * 2560, 2560
mk = lambda r, c: K.BF((torch.randn(r, c, device='cuda') * 0.5).to(torch.bfloat16), None)
shapes = [('qkv bf16', M, 1536, 512, True), ('to_out f32', M, 512, 512, False), ('ff1 bf16', M, 2752, 512, True), ('ff2 f32', M, 512, 1376, False),
          ('dgrad qkv f32', M, 512, 1536, False), ('dgrad ff1 f32', M, 512, 2752, False), ('dgrad ff2 bf16', M, 1376, 512, True),
          ('logits f32', M, 8192, 512, False), ('dgrad logits f32', M, 512, 8192, False)]
for name, m, nn, kk, obf in shapes:
    A, Bm = mk(m, kk), mk(nn, kk)
    row = []
    ref = None
    for var in variants:
        L.amdnuwa_set_tuning(0, var)
        out = K.gemm_nt(A, Bm, out_bf16=obf)
        o = (out.hi if obf else out).float()
        if ref is None:
            ref = o
        ok = bool(torch.equal(o, ref))
        for dbg in (0, 1, 2):
            L.amdnuwa_set_tuning(7, dbg)
            t = bench(lambda: K.gemm_nt(A, Bm, out_bf16=obf), 10)
            row.append(f'v{var} {["full", "no-st", "no-ml"][dbg]} {t * 1e6:6.1f}' + ('' if ok else ' MISMATCH'))
        L.amdnuwa_set_tuning(7, 0)
    L.amdnuwa_set_tuning(0, 0)
    tl = bench(lambda: torch.matmul(A.hi, Bm.hi.t()), 10)          # library yardstick (hipBLASLt / rocBLAS through torch), bf16 out
    row.append(f'torch.matmul {tl * 1e6:6.1f}')
    fl = 2.0 * m * nn * kk
    tiles = ((m + 255) // 256) * ((nn + 255) // 256)
    print(f'{name:18s} [{m}x{nn}x{kk}] tiles {tiles:6d} ({tiles / 256:.1f}/CU)  ' + ' | '.join(row) + f' | ideal mfma {fl / 2.5e15 * 1e6:6.1f} us')
