#!/usr/bin/env python
"""NUWASketch.generate on one MI355X (dim 512, 12 decoder layers, 5 x 16 x 16 video tokens, 2 sketch frames, SparseCross2DNA kernel 3):
per-token cost of the row-at-a-time decoder (decode.GuidedStepper; rows >= 1 replayed as a HIP graph, or launched eagerly) against
the reference's algorithm -- recompute the whole prefix, twice with guidance -- on the same training kernels.
  python tools/gen_sketch_bench.py [--batch 4] [--tokens 96]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuwa_pytorch_amd as A  # noqa: E402
from nuwa_pytorch_amd.decode import GuidedStepper  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--tokens', type=int, default=96)
    ap.add_argument('--cond-scale', type=float, default=2.)
    args = ap.parse_args()
    dev = 'cuda'
    torch.manual_seed(0)
    vae = A.VQGanVAE(dim=64, image_size=256, num_layers=4, vq_codebook_size=8192, use_vgg_and_gan=False)
    svae = A.VQGanVAE(dim=64, image_size=256, num_layers=4, vq_codebook_size=1024, use_vgg_and_gan=False)
    m = A.NUWASketch(vae=vae, sketch_vae=svae, dim=512, image_size=256, max_video_frames=5, sketch_max_video_frames=2, sketch_enc_depth=2,
                     dec_depth=12, dec_heads=8, dec_dim_head=64, cross_2dna_kernel_size=3, cross_2dna_dilation=2,
                     sparse_3dna_kernel_size=(5, 3, 3), sparse_3dna_dilation=(1, 2, 4)).to(dev).eval()
    b, N = args.batch, 5 * 256
    g = torch.Generator().manual_seed(1)
    sketch = torch.rand(b, 2, 3, 256, 256, generator=g).to(dev)
    ids = torch.randint(0, 8192, (b, N), generator=g).to(dev)
    with torch.no_grad():
        ctx, cmask = m.embed_sketch(sketch)
        rows = m.embed_video(ids[:, :args.tokens])
        res = {}
        for graph in (True, False):
            st = GuidedStepper(m, ctx, cmask, N, args.cond_scale, graph=graph)
            st(rows[:, 0])
            st(rows[:, 1])                       # (graph capture happens on row 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(2, args.tokens):
                st(rows[:, t])
            torch.cuda.synchronize()
            res[graph] = (time.perf_counter() - t0) / (args.tokens - 2)
            del st
        print(f'NUWASketch, b={b}, cond_scale={args.cond_scale}: cached row  graph {res[True] * 1e3:.2f} ms/token ({b / res[True]:.0f} tokens/s) | '
              f'eager {res[False] * 1e3:.2f} ms/token')
        tot = 0.
        for frac in (0.25, 0.5, 0.75):
            n = int(N * frac)

            def recompute():
                return m._guided_last_logits(ids[:, :n], ctx, cmask, args.cond_scale)
            recompute()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                recompute()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            tot += dt
            print(f'  recompute loop (reference algorithm, same kernels) at prefix {n}: {dt * 1e3:.1f} ms/token')
        print(f'  recompute average ~{tot / 3 * 1e3:.1f} ms/token; cached + graph speed-up ~{tot / 3 / res[True]:.0f}x')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vid = m.generate(sketch=sketch, num_frames=1, cond_scale=args.cond_scale)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'generate(num_frames=1): {tuple(vid.shape)} in {dt:.2f} s ({256 * b / dt:.0f} tokens/s incl. sampling + VAE decode)')


if __name__ == '__main__':
    main()
