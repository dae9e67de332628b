#!/usr/bin/env python
"""BASELINE cfg 5 at full size on one MI355X: NUWAVideoAudio (dim 512, class-default decoder depth 6, 8 heads x 64, 3DNA kernel 3 with
rel-pos bias, audio window 7), 10 frames x 16x16 video tokens + 32 audio tokens per frame (audio codebook 2048), 256 text tokens;
non-reversible dual decoder.  Times the whole training forward (text encoder included) + backward; prints tokens/s
(video + audio tokens).      python tools/cfg5_step.py [--batch 8] [--steps 5]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuwa_pytorch_amd as A  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--reversible', action='store_true', help='the class-default ReversibleDualModalityDecoder')
    args = ap.parse_args()
    dev = 'cuda'
    torch.manual_seed(0)
    vae = A.VQGanVAE(dim=64, image_size=256, num_layers=4, vq_codebook_size=8192, use_vgg_and_gan=False)
    m = A.NUWAVideoAudio(vae=vae, dim=512, image_size=256, num_audio_tokens=2048, num_audio_tokens_per_video_frame=32,
                         max_video_frames=10, text_max_seq_len=256, text_enc_depth=1, enc_reversible=True, dec_reversible=args.reversible).to(dev).train()
    b = args.batch
    g = torch.Generator().manual_seed(1)
    text = torch.randint(1, 49408, (b, 256), generator=g).to(dev)
    vid = torch.randint(0, 8192, (b, 10, 16, 16), generator=g).to(dev)
    aud = torch.randint(0, 2048, (b, 320), generator=g).to(dev)

    def step():
        m.zero_grad(set_to_none=True)
        loss = m(text=text, video=vid, audio=aud, return_loss=True, cond_dropout_prob=0.)
        loss.backward()
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f'cfg5 NUWAVideoAudio b={b}: {dt * 1e3:.1f} ms/step, {(2560 + 320) * b / dt:.0f} tokens/s (video + audio), loss {float(loss.detach()):.4f}')


if __name__ == '__main__':
    main()
