#!/usr/bin/env python
"""NUWAVideoAudio.generate at cfg-5 size (dim 512, 10 x 16 x 16 video tokens + 32 audio tokens per frame, default depth 6, reversible
dual decoder) on one MI355X: one frame pair (256 video + 32 audio tokens) with the per-layer caches (decode.DualGuidedStepper: one new
row per sampled token) and with the reference's algorithm (both decoders over the whole prefix, twice with guidance) on the same kernels.
  python tools/gen_va_bench.py [--batch 2] [--frames 1] [--plain]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuwa_pytorch_amd as A  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--frames', type=int, default=1)
    ap.add_argument('--cond-scale', type=float, default=2.)
    ap.add_argument('--plain', action='store_true', help='dec_reversible=False (DualModalityDecoder)')
    args = ap.parse_args()
    dev = 'cuda'
    torch.manual_seed(0)
    vae = A.VQGanVAE(dim=64, image_size=256, num_layers=4, vq_codebook_size=8192, use_vgg_and_gan=False)
    m = A.NUWAVideoAudio(vae=vae, dim=512, image_size=256, num_audio_tokens=2048, num_audio_tokens_per_video_frame=32, max_video_frames=10,
                         text_max_seq_len=256, text_enc_depth=1, enc_reversible=True, dec_reversible=not args.plain).to(dev).eval()
    text = torch.randint(1, 49408, (args.batch, 256), generator=torch.Generator().manual_seed(1)).to(dev)
    ntok = args.frames * (256 + 32)
    res = {}
    for cached in (True, 'eager', False):
        type(m).generate_use_cache, type(m).generate_use_graph = bool(cached), cached is True
        torch.manual_seed(0)
        m.generate(text=text, num_frames=1, cond_scale=args.cond_scale) if cached else None        # warm-up (weight caches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        video, audio = m.generate(text=text, num_frames=args.frames, cond_scale=args.cond_scale)
        torch.cuda.synchronize()
        res[cached] = time.perf_counter() - t0
    type(m).generate_use_cache = type(m).generate_use_graph = True
    kind = 'plain' if args.plain else 'reversible'
    print(f'cfg 5 ({kind} dual decoder), b={args.batch}, {args.frames} frame(s) = {ntok} tokens per sample, cond_scale={args.cond_scale}: '
          f'cached + HIP graphs {res[True]:.2f} s ({res[True] / ntok * 1e3:.1f} ms/token) | cached, eager launches {res["eager"]:.2f} s ({res["eager"] / ntok * 1e3:.1f} ms/token) | recompute loop {res[False]:.2f} s ({res[False] / ntok * 1e3:.1f} ms/token) | '
          f'{res[False] / res[True]:.1f}x  (video {tuple(video.shape)}, audio {tuple(audio.shape)})')


if __name__ == '__main__':
    main()
