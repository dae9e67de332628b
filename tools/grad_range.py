#!/usr/bin/env python
"""Range of the fp16 gradients of the 'bf16x3-fwd' backward: |S g| of every fp16 gradient tensor (kernels.G16) created during ONE bench step
(cfg 3, --depth layers, --batch samples), in creation order (logits end first), and the saturation counter.  fp16 ends at 65504.
    python tools/grad_range.py [--depth 24] [--batch 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import nuwa_pytorch_amd as A  # noqa: E402
from nuwa_pytorch_amd import kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--depth', type=int, default=24)
    ap.add_argument('--batch', type=int, default=8)
    args = ap.parse_args()
    c = dict(bench.CFGS['cfg3'], dec_depth=args.depth)
    nuwa = bench.build_model(c, 'cuda')
    g = torch.Generator().manual_seed(5)
    N = c['frames'] * c['fmap'] ** 2
    ids = torch.randint(0, c['codebook'], (args.batch, N), generator=g).cuda()
    ctx = torch.randn(args.batch, c['text_len'], c['dim'], generator=g).cuda()
    mask = (torch.rand(args.batch, c['text_len'], generator=g) > 0.1).cuda()
    A.set_precision('bf16x3-fwd')
    seen = []
    base = K.G16

    class Spy(base):
        def __new__(cls, t, s2):
            seen.append(t)
            return super().__new__(cls, t, s2)
    K.G16 = Spy
    K.f16_sat_count()
    bench.decoder_step(nuwa, ids, ctx, mask)
    torch.cuda.synchronize()
    am = [float(t.float().abs().max()) for t in seen]
    nf = sum(int((~torch.isfinite(t.float())).sum()) for t in seen)
    print(f'classes {"".join(sorted(K._BWD_F16))!r}, depth {args.depth}, batch {args.batch}: {len(seen)} fp16 gradient tensors, non-finite entries {nf}, '
          f'saturation counter {K.f16_sat_count()}')
    print('|S g| max per tensor, creation order:', ' '.join(f'{a:.0f}' for a in am))
    print(f'largest {max(am):.0f} = 2^{torch.log2(torch.tensor(max(am))).item():.1f} (fp16 max 2^16)')


if __name__ == '__main__':
    main()
