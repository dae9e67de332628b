#!/usr/bin/env python
"""GEMM micro-benchmark on the MI355X: every GEMM shape of the cfg-3 decoder step (per-GPU batch b),
under each tuning variant of libamdnuwa (NT: register-staged / direct-to-LDS BK64 / BK32; TN: split-K
policies).  Prints TFLOP/s per shape + variant and checks the variants against variant 0.
    python tools/gemm_bench.py [--batch 8] [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402


def bench(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    L = _lib.lib()
    dev = 'cuda'
    b, n, D, inner, FP, C, T = args.batch, 2560, 512, 512, 1376, 8192, 256
    M = b * n
    torch.manual_seed(0)
    mk = lambda r, c: K.BF((torch.randn(r, c, device=dev) * 0.5).to(torch.bfloat16), None)
    nt_shapes = [('qkv  (shift, bf16 out)', M, 3 * inner, D, True, True), ('to_out (fp32+bias)', M, D, inner, False, False),
                 ('q (bf16 out)', M, inner, D, True, False), ('ff1 (shift, bf16 out)', M, 2 * FP, D, True, True),
                 ('ff2 (fp32)', M, D, FP, False, False), ('dgrad qkv (fp32)', M, D, 3 * inner, False, False),
                 ('dgrad ff1 (fp32)', M, D, 2 * FP, False, False), ('dgrad ff2 (bf16)', M, FP, D, True, False),
                 ('logits (fp32)', M, C, D, False, False), ('dgrad logits (fp32)', M, D, C, False, False)]
    print(f'== NT  (M = {M}) ==')
    for name, m, nn, kk, obf, sh in nt_shapes:
        A, Bm = mk(m, kk), mk(nn, kk)
        ref = None
        row = []
        for var in (0, 5, 2, 7):      # 0 = auto, 5 = register-staged, 2 = glds BK32, 7 = the 256x256 ring for every size
            L.amdnuwa_set_tuning(0, var)
            out = K.gemm_nt(A, Bm, out_bf16=obf, shift=(n, 16) if sh else None)
            o = out.hi.float() if obf else out
            if ref is None:
                ref = o.clone()
                ok = True
            else:
                ok = bool(torch.equal(o, ref)) or float((o - ref).abs().max() / ref.abs().max()) < 1e-5
            t = bench(lambda: K.gemm_nt(A, Bm, out_bf16=obf, shift=(n, 16) if sh else None), args.iters)
            row.append(f'v{var}: {2.0 * m * nn * kk / t / 1e12:7.1f} TF/s ({t * 1e6:7.1f} us){"" if ok else " MISMATCH"}')
        print(f'{name:26s} [{m}x{nn}x{kk}]  ' + ' | '.join(row))
    L.amdnuwa_set_tuning(0, 0)
    print(f'== TN  (rows = {M}) ==')
    tn_shapes = [('dW qkv (shift)', 3 * inner, D, True), ('dW out', D, inner, False), ('dW ff1 half (shift)', 1365, D, True),
                 ('dW ff2', D, 1365, False), ('dW logits', C, D, False)]
    for name, n1, n2, sh in tn_shapes:
        ld1, ld2 = (n1 + 31) // 32 * 32, (n2 + 31) // 32 * 32
        A, Bm = mk(M, ld1), mk(M, ld2)
        out = torch.empty(n1, n2, device=dev)
        row = []
        ref = None
        for variant, target, minrows in ((0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0), (3, 256, 1024), (2, 1024, 1024)):
            L.amdnuwa_set_tuning(6, variant)
            L.amdnuwa_set_tuning(1, target)
            L.amdnuwa_set_tuning(2, minrows)
            f = lambda: K.gemm_tn(K.view(A, cols=slice(0, n1)), K.view(Bm, cols=slice(0, n2)), out, shift=(n, 16) if sh else None, N1=n1, N2=n2)
            f()
            if ref is None:
                ref = out.clone()
            err = float((out - ref).abs().max() / ref.abs().max())
            t = bench(f, args.iters)
            row.append(f'v{variant}/wg{target}/r{minrows}: {2.0 * M * n1 * n2 / t / 1e12:6.1f} TF/s ({t * 1e6:6.1f} us){"" if err < 1e-4 else " MISMATCH"}')
        print(f'{name:22s} [{n1}x{n2}]  ' + ' | '.join(row))
    L.amdnuwa_set_tuning(1, 0)
    L.amdnuwa_set_tuning(2, 0)
    L.amdnuwa_set_tuning(6, 0)


if __name__ == '__main__':
    main()
