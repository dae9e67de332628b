#!/usr/bin/env python
"""Ablation probes of the attention kernels at the cfg-3 geometry (timing only: the probe bits produce garbage results).
Sparse3DNA forward (tuning key 9: phase skips) on one- and two-row tiles, Sparse3DNA backward (key 17), cross-attention forward and
backward (key 18; key 10 bit 4 = no dS / P' stores)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from attn_bench import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--iters', type=int, default=8)
    ap.add_argument('--dil', type=int, default=2)
    args = ap.parse_args()
    K.set_precision('bf16')
    L = _lib.lib()
    dev = 'cuda'
    b, n, heads, dh, T = args.batch, 2560, 8, 64, 256
    inner = heads * dh
    torch.manual_seed(0)
    qkv = K.BF((torch.randn(b * n, 3 * inner, device=dev)).to(torch.bfloat16), None)
    qkv16 = K.BF(qkv.hi, None, qkv.hi.float().half())
    do = K.BF((torch.randn(b * n, inner, device=dev)).to(torch.bfloat16), None)
    wth = (torch.randn(heads, heads, device=dev) * 0.3 + torch.eye(heads, device=dev)).contiguous()
    g = K.s3_geom(b, n, (10, 16, 16), (5, 3, 3), (args.dil,) * 3, heads, dh)
    print(f'== Sparse3DNA forward (fp16 operands), b={b}, dilation {args.dil}: tuning key 9 phase skips ==')
    for rws in (1, 2):
        L.amdnuwa_set_tuning(16, rws)
        row = []
        for nm, bits in (('full', 0), ('no scores', 1), ('no softmax+mix', 2), ('no apply', 4), ('scores only', 6), ('apply only', 3), ('nothing', 7)):
            L.amdnuwa_set_tuning(9, bits)
            row.append(f'{nm} {bench(lambda: K.sparse3dna_fwd(g, qkv16, wth), args.iters) * 1e6:7.1f}')
        L.amdnuwa_set_tuning(9, 0)
        print(f'rows {rws}: ' + ' | '.join(row))
    L.amdnuwa_set_tuning(16, 0)
    print('== Sparse3DNA backward: tuning key 17 (bit 0 no score sweeps, 1 no workspace stores, 2 no dq apply, 3 no dW_th; 4 no coefficient gathers, 5 no q / dO row fetch) ==')
    row = []
    for nm, bits in (('full', 0), ('q: no sweeps', 1), ('q: no ws stores', 2), ('q: no apply', 4), ('q: no dWth', 8), ('q: none of them', 15),
                     ('kv: no coef gathers', 16), ('kv: no row fetch', 32), ('kv: neither', 48), ('all off', 63)):
        L.amdnuwa_set_tuning(17, bits)
        row.append(f'{nm} {bench(lambda: K.sparse3dna_bwd(g, qkv, wth, do), args.iters) * 1e6:7.1f}')
    L.amdnuwa_set_tuning(17, 0)
    print(' | '.join(row))
    row = []
    for nm, v in (('fused item pass', 0), ('three separate item passes', 1), ('fused item pass', 0), ('three separate item passes', 1)):    # tuning key 19
        L.amdnuwa_set_tuning(19, v)
        row.append(f'{nm} {bench(lambda: K.sparse3dna_bwd(g, qkv, wth, do), args.iters) * 1e6:7.1f}')
    L.amdnuwa_set_tuning(19, 0)
    print('bwd, query side item passes (A/B/A/B): ' + ' | '.join(row))


if __name__ == '__main__':
    main()
