#!/usr/bin/env python
"""NUWA.generate at BASELINE cfg 3 on one MI355X: per-token cost of (i) the key/value-cached step replayed as a HIP graph,
(ii) the same step launched eagerly, (iii) the reference's algorithm -- recompute the whole prefix twice -- on the same
kernels, measured at prefix lengths 1/4, 1/2 and 3/4 of the video (its cost grows with the prefix; the cached step's does not).
  python tools/gen_bench.py [--batch 4] [--tokens 128]"""
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
    ap.add_argument('--tokens', type=int, default=128)
    ap.add_argument('--cond-scale', type=float, default=2.)
    args = ap.parse_args()
    dev = 'cuda'
    torch.manual_seed(0)
    vae = A.VQGanVAE(dim=64, image_size=256, num_layers=4, vq_codebook_size=8192, use_vgg_and_gan=False)
    nuwa = A.NUWA(vae=vae, dim=512, max_video_frames=10, text_max_seq_len=256, text_enc_depth=6, enc_reversible=True, dec_depth=24,
                  dec_heads=8, dec_dim_head=64, sparse_3dna_kernel_size=(5, 3, 3), sparse_3dna_dilation=(1, 2, 4),
                  shift_video_tokens=True).to(dev).eval()
    b, N = args.batch, 2560
    g = torch.Generator().manual_seed(1)
    text = torch.randint(1, 49408, (b, 256), generator=g).to(dev)
    ids = torch.randint(0, 8192, (b, N), generator=g).to(dev)
    with torch.no_grad():
        mask = text != 0
        emb = nuwa.embed_text(text, mask=mask)
        rows = nuwa.embed_video(ids[:, :args.tokens])
        res = {}
        for graph in (True, False):
            st = GuidedStepper(nuwa, emb, mask, N, args.cond_scale, graph=graph)
            st(rows[:, 0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(1, args.tokens):
                st(rows[:, t])
            torch.cuda.synchronize()
            res[graph] = (time.perf_counter() - t0) / (args.tokens - 1)
            del st
        print(f'cfg 3, b={b}, cond_scale={args.cond_scale}: cached step  graph {res[True] * 1e3:.2f} ms/token '
              f'({b / res[True]:.0f} tokens/s, {N * res[True]:.1f} s per {N}-token video) | eager {res[False] * 1e3:.2f} ms/token')
        tot = 0.
        for frac in (0.25, 0.5, 0.75):
            n = int(N * frac)
            x = nuwa.embed_video(ids[:, :n])

            def recompute():
                hidden = nuwa.decode_hidden(x, emb, mask)
                lg = nuwa._final(hidden)
                if args.cond_scale != 1:
                    un = nuwa._final(nuwa.decode_hidden(nuwa.video_transformer.norm(hidden), emb, torch.zeros_like(mask)))
                    lg = un + (lg - un) * args.cond_scale
                return lg[:, -1]
            recompute()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                recompute()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            tot += dt
            print(f'  recompute loop (reference algorithm, same kernels) at prefix {n}: {dt * 1e3:.1f} ms/token')
        avg = tot / 3
        print(f'  recompute average ~{avg * 1e3:.1f} ms/token -> ~{N * avg:.0f} s per video; cached+graph speed-up ~{avg / res[True]:.0f}x')
    # end-to-end generate() of a short clip (2 frames = 512 tokens) incl. sampling and the VAE decode
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vid = nuwa.generate(text=text, num_frames=2, cond_scale=args.cond_scale)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'generate(num_frames=2): {tuple(vid.shape)} in {dt:.2f} s ({512 * b / dt:.0f} tokens/s incl. sampling + VAE decode)')


if __name__ == '__main__':
    main()
