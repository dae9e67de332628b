#!/usr/bin/env python
"""The two-MFMA form (fp16 activation x fp16 hi + lo weight, gemm_nt_256x3_kernel<EPI, X2>) next to the three-MFMA hi + lo ring on the shapes
that take it in the 'bf16x3-fwd' forward: full / epilogue stores skipped (tuning 7 = 1) / main loop skipped (tuning 7 = 2)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from gemm_bench import bench  # noqa: E402

L = _lib.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M = b * 2560


def mk(r, c):
    t = torch.randn(r, c, device='cuda') * 0.5
    hi = t.to(torch.bfloat16)
    return t, K.BF(hi, (t - hi.float()).to(torch.bfloat16))


K.set_precision('bf16x3-fwd')
for name, m, nn, kk, obf in (('to_out f32', M, 512, 512, False), ('xattn q bf16 + f16', M, 512, 512, True), ('logits f32', M, 8192, 512, False)):
    ta, A = mk(m, kk)
    tb, Bm = mk(nn, kk)
    a16, wp = ta.half(), K.f16_pair(tb)
    calls = {'x3': (lambda: K.gemm_nt(A, Bm, out_bf16=obf, out_f16=obf)), 'x2': (lambda: K.gemm_nt_f16x2(a16, wp, out_bf16=obf, copy_f16=obf))}
    fl = 2.0 * m * nn * kk
    for form, call in calls.items():
        row = []
        for dbg in (0, 1, 2):
            L.amdnuwa_set_tuning(7, dbg)
            t = bench(call, 10)
            row.append(f'{["full", "no-st", "no-ml"][dbg]} {t * 1e6:7.1f}')
            if dbg == 0:
                t_full = t
        L.amdnuwa_set_tuning(7, 0)
        nm = 3 if form == 'x3' else 2
        print(f'{name:18s} {form} [{m}x{nn}x{kk}]  ' + ' | '.join(row) + f' | issued {nm * fl / t_full / 1e12:6.0f} TF/s | ideal {nm}x mfma {nm * fl / 2.5e15 * 1e6:6.1f} us')
