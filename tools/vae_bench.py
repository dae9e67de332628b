#!/usr/bin/env python
"""Frozen-tokenizer micro-benchmark on the MI355X: VQGanVAE.get_video_indices at the cfg-3 VAE (dim 64, 256x256 frames,
4 layers, codebook 8192 x 256) through libamdnuwa's exact-fp32 kernels; per-stage times and fp32 TFLOP/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuwa_pytorch_amd as A  # noqa: E402
from nuwa_pytorch_amd import kernels as K  # noqa: E402


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
    ap.add_argument('--frames', type=int, default=80)
    ap.add_argument('--iters', type=int, default=5)
    args = ap.parse_args()
    torch.manual_seed(0)
    vae = A.VQGanVAE(dim=64, image_size=256, num_layers=4, vq_codebook_size=8192, use_vgg_and_gan=False).eval().cuda()
    video = torch.rand(args.frames // 10, 10, 3, 256, 256, device='cuda')
    t = bench(lambda: vae.get_video_indices(video), args.iters)
    print(f'get_video_indices: {args.frames} frames  {t * 1e3:8.2f} ms  {args.frames / t:9.1f} frames/s  {args.frames * 256 / t:10.0f} tokens/s')
    x = video.reshape(-1, 3, 256, 256)
    with torch.no_grad():
        for i, enc in enumerate(vae.encoders):
            tt = bench(lambda: vae._hip_module(enc, x), args.iters)
            fl = 0
            convs = [m for m in enc.modules() if isinstance(m, torch.nn.Conv2d)]
            y = vae._hip_module(enc, x)
            for c in convs:
                ho = y.shape[-1]
                fl += 2 * x.shape[0] * c.out_channels * c.in_channels * c.kernel_size[0] * c.kernel_size[1] * ho * ho
            print(f'  encoders[{i}] {type(enc).__name__:16s} {tuple(x.shape)} -> {tuple(y.shape)}  {tt * 1e3:8.3f} ms  {fl / tt / 1e12:6.1f} TF/s (conv flops)')
            x = y
        rows = K.conv2d_fwd(x, vae.vq.project_in.weight[:, :, None, None], vae.vq.project_in.bias).permute(0, 2, 3, 1).reshape(-1, 256)
        tt = bench(lambda: K.vq_argmax(rows, vae.vq.embed), args.iters)
        print(f'  vq_argmax {tuple(rows.shape)} x 8192 codes  {tt * 1e3:8.3f} ms  {2 * rows.shape[0] * 8192 * 256 / tt / 1e12:6.1f} TF/s')


if __name__ == '__main__':
    main()
