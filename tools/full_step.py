#!/usr/bin/env python
"""The WHOLE reference training step at BASELINE cfg 3 on one MI355X: NUWA.forward(text ids, raw video frames, return_loss=True)
= text encoder (depth 6, reversible, rotary) + frozen VAE tokenizer (10 frames of 256x256 per sample) + 24-layer 3DNA decoder + CE,
then backward.  Every stage runs through libamdnuwa.   python tools/full_step.py [--batch 16] [--steps 3]"""
import argparse
import os
import sys
import time

# the caching allocator with expandable segments: at b = 128 the step peaks at 246 GiB of the 268 GiB card, and with fixed-size segments the
# allocator spends 80 ms per step retrying (800 vs 719 ms, profiles/r05_full_step.txt); set before torch is imported
os.environ.setdefault('PYTORCH_HIP_ALLOC_CONF', 'expandable_segments:True')
os.environ.setdefault('PYTORCH_CUDA_ALLOC_CONF', 'expandable_segments:True')

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nuwa_pytorch_amd as A  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--optimizer', action='store_true', help='add the fused clip(0.5) + AdamW step of the trainer (row f2)')
    args = ap.parse_args()
    dev = 'cuda'
    # from b = 112 the fused to_logits + cross entropy (no fp32 logits: 9 GiB less at the peak) unless the environment says otherwise: at 244 of
    # 288 GiB the unfused step ran 708-919 ms depending on the allocator's retries, fused 701-713 ms (profiles/r05o_full_step_auto_ce.txt)
    from nuwa_pytorch_amd import ops
    if 'AMDNUWA_FUSE_LINEAR_CE_X3' not in os.environ and args.batch >= 112:
        ops.FUSE_LINEAR_CE_X3 = True
    torch.manual_seed(0)
    vae = A.VQGanVAE(dim=64, image_size=256, num_layers=4, vq_codebook_size=8192, use_vgg_and_gan=False)
    nuwa = A.NUWA(vae=vae, dim=512, max_video_frames=10, text_max_seq_len=256, text_enc_depth=6, enc_reversible=True, dec_depth=24,
                  dec_heads=8, dec_dim_head=64, sparse_3dna_kernel_size=(5, 3, 3), sparse_3dna_dilation=(1, 2, 4),
                  shift_video_tokens=True).to(dev).train()
    opt = None
    if args.optimizer:
        from nuwa_pytorch_amd.optimizer import get_optimizer
        opt = get_optimizer(nuwa.parameters(), lr=3e-4, wd=0.01, filter_by_requires_grad=True)
    b = args.batch
    g = torch.Generator().manual_seed(1)
    text = torch.randint(1, 49408, (b, 256), generator=g)
    text[:, -32:] = 0
    text = text.to(dev)
    video = torch.rand(b, 10, 3, 256, 256, generator=g).to(dev)

    def step():
        nuwa.zero_grad(set_to_none=True)
        loss = nuwa(text=text, video=video, return_loss=True, cond_dropout_prob=0.2)
        loss.backward()
        if opt is not None:
            opt.step(max_grad_norm=0.5)
        return loss
    for _ in range(args.warmup):       # (the caching allocator still maps new segments in the second step of a run at 235 GiB: one warm-up step left 663 ... 830 ms in the timed ones)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    times = []
    for _ in range(args.steps):
        t1 = time.perf_counter()
        loss = step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ce_form = 'fused' if ops.FUSE_LINEAR_CE_X3 is True else ('auto' if ops.FUSE_LINEAR_CE_X3 else 'unfused')
    print(f'[{A.get_precision()}] full NUWA step (text encoder + VAE tokenizer + decoder fwd/bwd{" + clip + AdamW" if opt else ""}), cfg 3, b={b}, {ce_form} logits + CE: {dt * 1e3:.1f} ms/step, '
          f'{2560 * b / dt:.0f} video-tokens/s, loss {float(loss.detach()):.4f}, peak {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB, reserved '
          f'{torch.cuda.max_memory_reserved() / 2 ** 30:.1f} GiB, allocator retries {torch.cuda.memory_stats().get("num_alloc_retries", 0)}, per step ' + ' / '.join(f'{t:.0f}' for t in times) + ' ms')


if __name__ == '__main__':
    main()
