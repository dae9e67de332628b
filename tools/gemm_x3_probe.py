#!/usr/bin/env python
"""The bf16x3 256x256 ring (forward GEMMs of the 'bf16x3-fwd' mode) per decoder shape: full / epilogue stores skipped (tuning 7 = 1) /
main loop skipped (tuning 7 = 2), next to the first-generation 128x128 x3 kernel (tuning 13 = 1).  `issued` = 3 x 2MNK / time."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from gemm_bench import bench  # noqa: E402

L = _lib.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = b * 2560


def mk(r, c):
    t = torch.randn(r, c, device='cuda') * 0.5
    hi = t.to(torch.bfloat16)
    return K.BF(hi, (t - hi.float()).to(torch.bfloat16))


shapes = [('qkv hi+lo', M, 1536, 512, True, False), ('xattn q hi+lo', M, 512, 512, True, False), ('to_out f32', M, 512, 512, False, False),
          ('ff1 hi+lo', M, 2752, 512, True, False), ('ff1 + gate', M, 2752, 512, True, True), ('ff2 f32', M, 512, 1376, False, False),
          ('logits f32', M, 8192, 512, False, False)]
K.set_precision('bf16x3-fwd')
for name, m, nn, kk, obf, gate in shapes:
    A, Bm = mk(m, kk), mk(nn, kk)
    gg = K.empty_bf((m, nn // 2), 'cuda', lo=True) if gate else None
    call = lambda: K.gemm_nt(A, Bm, out_bf16=obf, geglu_out=gg)
    row = []
    for dbg in (0, 1, 2):
        L.amdnuwa_set_tuning(7, dbg)
        t = bench(call, 10)
        row.append(f'{["full", "no-st", "no-ml"][dbg]} {t * 1e6:7.1f}')
        if dbg == 0:
            t_full = t
    L.amdnuwa_set_tuning(7, 0)
    L.amdnuwa_set_tuning(13, 1)
    t_old = bench(call, 5)
    L.amdnuwa_set_tuning(13, 0)
    for sk in (15, 40, 80):                      # start-phase skew (tuning key 14), ~0.25 us units per phase step
        L.amdnuwa_set_tuning(14, sk)
        row.append(f'skew{sk} {bench(call, 10) * 1e6:7.1f}')
    L.amdnuwa_set_tuning(14, 0)
    fl = 2.0 * m * nn * kk
    print(f'{name:14s} [{m}x{nn}x{kk}]  ' + ' | '.join(row) + f' | 128x128 kernel {t_old * 1e6:7.1f} | issued {3 * fl / t_full / 1e12:6.0f} TF/s | ideal 3x mfma {3 * fl / 2.5e15 * 1e6:6.1f} us')
