#!/usr/bin/env python
"""Cross-attention cores at the cfg-3 geometry (per-GPU batch b, n = 2560, T = 256): the third design (amdnuwa_xattn6_*) beside the
second (xattn_pack + xattn4 forward, xattn3 backward + batched TN products), A/B/A/B in one process."""
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
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--T', type=int, default=256)
    args = ap.parse_args()
    dev = 'cuda'
    b, n, heads, dh, T = args.batch, 2560, 8, 64, args.T
    inner = heads * dh
    torch.manual_seed(0)
    g = K.x_geom(b, n, T, heads, dh)
    q16 = torch.randn(b * n, inner, device=dev).half()
    kv16 = torch.randn(b * T, 2 * inner, device=dev).half()
    q = K.BF(q16.to(torch.bfloat16), None, q16)
    kv = K.BF(kv16.to(torch.bfloat16), None, kv16)
    nk, nv = torch.randn(heads, dh, device=dev), torch.randn(heads, dh, device=dev)
    wth = (torch.randn(heads, heads, device=dev) * 0.3 + torch.eye(heads, device=dev)).contiguous()
    mask = (torch.rand(b, T, device=dev) > 0.2).to(torch.uint8)
    fl = 4 * heads * n * (T + 1) * dh * b
    print(f'== cross-attention forward, b={b}, n={n}, T={T} (MFMA-useful = QK^T + P\'V) ==')
    pk6 = K.xattn6_pack(g, kv16, mask)
    rows = []
    for rnd in range(2):
        t = bench(lambda: K.xattn6_pack(g, kv16, mask, out=pk6), args.iters)
        rows.append(f'pack6 {t:7.1f}')
        t = bench(lambda: K.xattn6_fwd(g, q16, pk6, nk, nv, wth, o_f16=True), args.iters)
        rows.append(f'xattn6_fwd {t:7.1f} us ({fl / t / 1e6:6.1f} TF/s)')
        if T + 1 <= 287:
            t = bench(lambda: K.xattn_pack(g, kv, nk, nv, mask, lean=True), args.iters)
            rows.append(f'pack(lean) {t:7.1f}')
            pk = K.xattn_pack(g, kv, nk, nv, mask, lean=True)
            t = bench(lambda: K.xattn2_fwd_f16(g, q, pk, wth, o_f16=True), args.iters)
            rows.append(f'xattn4_fwd {t:7.1f} us ({fl / t / 1e6:6.1f} TF/s)')
    print(' | '.join(rows))
    o6, st6 = K.xattn6_fwd(g, q16, pk6, nk, nv, wth, o_f16=True)
    if T + 1 <= 287:
        o4, st4 = K.xattn2_fwd_f16(g, q, pk, wth, o_f16=True)
        d = (o6.f16.float() - o4.f16.float()).abs().max().item() / o4.f16.float().abs().max().item()
        print(f'max |o6 - o4| / max |o4| = {d:.2e}')


if __name__ == '__main__' and not (len(sys.argv) > 1 and sys.argv[1] == 'bwd'):
    main()


def bwd_main():
    """python tools/xattn6_bench.py bwd [--batch b]: query side of the backward, third design against the second, A/B/A/B"""
    b = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 128
    dev = 'cuda'
    n, heads, dh, T = 2560, 8, 64, 256
    inner = heads * dh
    torch.manual_seed(0)
    g = K.x_geom(b, n, T, heads, dh)
    q = K.BF(torch.randn(b * n, inner, device=dev).to(torch.bfloat16), None)
    do = K.BF(torch.randn(b * n, inner, device=dev).to(torch.bfloat16), None)
    kv = K.BF(torch.randn(b * T, 2 * inner, device=dev).to(torch.bfloat16), None)
    nk, nv = torch.randn(heads, dh, device=dev), torch.randn(heads, dh, device=dev)
    wth = (torch.randn(heads, heads, device=dev) * 0.3 + torch.eye(heads, device=dev)).contiguous()
    mask = (torch.rand(b, T, device=dev) > 0.2).to(torch.uint8)
    pk6 = K.xattn6_pack(g, kv.hi, mask)
    o, stats = K.xattn6_fwd(g, q.hi, pk6, nk, nv, wth, lo=False)
    pkb = K.xattn6_pack_bwd(g, kv.hi, nk, nv, mask)
    pko = K.xattn_pack(g, kv, nk, nv, mask)
    rows = []
    for rnd in range(2):
        rows.append(f'pack_bwd6 {bench(lambda: K.xattn6_pack_bwd(g, kv.hi, nk, nv, mask), 10):6.1f}')
        rows.append(f'xattn6_bwd {bench(lambda: K.xattn6_bwd(g, q, do, pkb, wth, stats), 10):7.1f}')
        rows.append(f'xattn3_bwd {bench(lambda: K.xattn2_bwd(g, q, do, pko, wth, stats, chunk_major=True), 10):7.1f}')
    print(f'== cross-attention backward, query side, b={b} (incl. the colsum of the dW_th partials) ==')
    print(' | '.join(rows))
    dq, dS, Pm, dwth = K.xattn6_bwd(g, q, do, pkb, wth, stats)
    rows = []
    rows.append(f'kv grads (2 batched TN) {bench(lambda: K.xattn_kv_grads(g, dS, Pm, q, do), 10):7.1f}')
    print(' | '.join(rows))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'bwd':
    bwd_main()
