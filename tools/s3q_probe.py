#!/usr/bin/env python
"""Phase probes of the Sparse3DNA backward, query side, fp16-gradient form (timing only: probe bits of tuning key 17 give garbage results).
bits: 1 no score sweeps, 2 no workspace stores, 4 no dq apply, 8 no dW_th FMAs, 64 no softmax, 128 no item pass, 256 no ds pass, 512 no pack pass,
1024 no <bos> partials, 2048 no table init; 16 / 32: key side (coefficient gathers / row fetch)."""
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
    ap.add_argument('--dils', type=str, default='1,2,4')
    args = ap.parse_args()
    K.set_precision('bf16x3-fwd')
    L = _lib.lib()
    dev = 'cuda'
    b, n, heads, dh = args.batch, 2560, 8, 64
    inner = heads * dh
    torch.manual_seed(0)
    qkv16 = torch.randn(b * n, 3 * inner, device=dev).half()
    do16 = torch.randn(b * n, inner, device=dev).half()
    wth = (torch.randn(heads, heads, device=dev) * 0.3 + torch.eye(heads, device=dev)).contiguous()
    s2 = torch.tensor([1.0, 1.0], device=dev)
    for dil in [int(x) for x in args.dils.split(',')]:
        g = K.s3_geom(b, n, (10, 16, 16), (5, 3, 3), (dil,) * 3, heads, dh)
        row = []
        for nm, bits in (('full', 0), ('no sweeps', 1), ('no ws stores', 2), ('no apply', 4), ('no softmax', 64), ('no item pass', 128), ('no ds pass', 256),
                         ('no pack pass', 512), ('no bos partials', 1024), ('no table init', 2048), ('q: tables only (no sweeps/apply)', 5),
                         ('q: sweeps+apply only', 64 + 128 + 256 + 512 + 1024), ('q: nothing', 1 + 4 + 64 + 128 + 256 + 512 + 1024 + 2048), ('kv: neither', 48), ('full', 0)):
            L.amdnuwa_set_tuning(17, bits)
            row.append(f'{nm} {bench(lambda: K.sparse3dna_bwd16(g, qkv16, wth, do16, s2), args.iters) * 1e6:7.1f}')
        L.amdnuwa_set_tuning(17, 0)
        print(f'dilation {dil}: ' + ' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
