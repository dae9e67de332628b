#!/usr/bin/env python
"""Attention-core micro-benchmark on the MI355X at the cfg-3 geometry (per-GPU batch b): Sparse3DNA fwd/bwd for
each dilation and staging variant, cross-attention fwd/bwd; reports time and achieved HBM GB/s against the
ALGORITHMIC bytes (fwd: read q,k,v + write o; bwd: read q,k,v,dO + write dq,dk,dv)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402


def bench(fn, iters):
    for _ in range(2):
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
    ap.add_argument('--iters', type=int, default=10)
    args = ap.parse_args()
    L = _lib.lib()
    dev = 'cuda'
    b, n, heads, dh, T = args.batch, 2560, 8, 64, 256
    inner = heads * dh
    torch.manual_seed(0)
    qkv = K.BF((torch.randn(b * n, 3 * inner, device=dev)).to(torch.bfloat16), None)
    do = K.BF((torch.randn(b * n, inner, device=dev)).to(torch.bfloat16), None)
    wth = (torch.randn(heads, heads, device=dev) * 0.3 + torch.eye(heads, device=dev)).contiguous()
    by_f, by_b = 4 * b * n * inner * 2, 8 * b * n * inner * 2
    print(f'== Sparse3DNA core, b={b}, 10x16x16, kernel (5,3,3), 8 heads x 64 ==')
    for dil in (1, 2, 4):
        g = K.s3_geom(b, n, (10, 16, 16), (5, 3, 3), (dil, dil, dil), heads, dh)
        row = []
        for v in (0, 1):
            L.amdnuwa_set_tuning(3, v)
            t = bench(lambda: K.sparse3dna_fwd(g, qkv, wth), args.iters)
            row.append(f'fwd[{"mfma" if v == 0 else "dot2"}] {t * 1e6:7.1f} us ({by_f / t / 1e9:6.0f} GB/s)')
        L.amdnuwa_set_tuning(3, 0)
        qkv16 = K.BF(qkv.hi, None, qkv.hi.float().half())
        for rws in (1, 2, 4):                                  # tuning key 16: query rows per workgroup of the MFMA forward
            L.amdnuwa_set_tuning(16, rws)
            t = bench(lambda: K.sparse3dna_fwd(g, qkv, wth), args.iters)
            t16 = bench(lambda: K.sparse3dna_fwd(g, qkv16, wth), args.iters)
            row.append(f'fwd[rows {rws}] bf16 {t * 1e6:7.1f} / f16 {t16 * 1e6:7.1f} us')
        L.amdnuwa_set_tuning(16, 0)
        for v in (0, 1):
            L.amdnuwa_set_tuning(4, v)
            t = bench(lambda: K.sparse3dna_bwd(g, qkv, wth, do), args.iters)
            row.append(f'bwd[{"mfma-q" if v == 0 else "dot2"}] {t * 1e6:7.1f} us ({by_b / t / 1e9:6.0f} GB/s)')
        L.amdnuwa_set_tuning(4, 0)
        print(f'dilation {dil}: ' + ' | '.join(row))
    print(f'== cross-attention core, b={b}, n={n}, T={T} ==')
    g = K.x_geom(b, n, T, heads, dh)
    q = K.BF(torch.randn(b * n, inner, device=dev).to(torch.bfloat16), None)
    kv = K.BF(torch.randn(b * T, 2 * inner, device=dev).to(torch.bfloat16), None)
    nk = torch.randn(heads, dh, device=dev)
    mask = (torch.rand(b, T, device=dev) > 0.2).to(torch.uint8)
    t = bench(lambda: K.xattn_pack(g, kv, nk, nk, mask), args.iters)
    print(f'pack        {t * 1e6:7.1f} us')
    pk = K.xattn_pack(g, kv, nk, nk, mask)
    fl = (4 * heads * n * g.JP * dh) * b
    for v in (0, 1):
        L.amdnuwa_set_tuning(5, v)
        t = bench(lambda: K.xattn_fwd(g, q, pk, wth), args.iters)
        print(f'fwd[{"unrolled" if v == 0 else "generic"}] {t * 1e6:7.1f} us ({fl / t / 1e12:6.1f} TF/s MFMA-useful)')
    L.amdnuwa_set_tuning(5, 0)
    o, P, Pm = K.xattn_fwd(g, q, pk, wth)
    t = bench(lambda: K.xattn_bwd(g, do, pk, wth, P), args.iters)
    print(f'bwd_q       {t * 1e6:7.1f} us')
    dq, dS, dw = K.xattn_bwd(g, do, pk, wth, P)
    for v in (0, 1):
        L.amdnuwa_set_tuning(6, v)
        t = bench(lambda: K.xattn_kv_grads(g, dS, Pm, q, do), args.iters)
        print(f'kv grads (2 batched TN, tn variant {v}) {t * 1e6:7.1f} us')
    L.amdnuwa_set_tuning(6, 0)
    if K.xattn2_supported(g, q):       # second design (xattn4 forward, xattn3 backward) beside the third (tools/xattn6_bench.py times both A/B)
        t = bench(lambda: K.xattn2_fwd(g, q, pk, wth), args.iters)
        o2, stats = K.xattn2_fwd(g, q, pk, wth)
        t2 = bench(lambda: K.xattn2_bwd(g, q, do, pk, wth, stats), args.iters)
        print(f'xattn4 fwd {t * 1e6:7.1f} us ({fl / t / 1e12:6.1f} TF/s MFMA-useful)  xattn3 bwd_q {t2 * 1e6:7.1f} us')


if __name__ == '__main__':
    main()
