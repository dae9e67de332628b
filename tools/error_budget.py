#!/usr/bin/env python
"""Error budget of the cfg-3 logits by operand rounding -- runs on the CPU, no GPU needed.

The north star wants the logits within 1e-3 (max-abs / max-abs) of the fp32 reference.  The all-bf16 mode measures 7.9e-3 on the
GPU; this tool answers "which roundings cost how much, and would fp16 operands be enough?" WITHOUT writing a kernel first: it runs
the oracle's arithmetic (oracle/nuwa_oracle.py, fp32) with a rounding injected at every place where the HIP kernels hand an
operand to an MFMA or store an activation in 16 bits:

    class   what is rounded                                                   where (kernel side)
    h       LayerNorm outputs (the A operand of every first GEMM of a block)  ln_fwd / ln_post_pre stores
    w       weight copies (B operands)                                        WeightCache casts
    a       projection outputs feeding an attention core (q, k, v; context)  qkv / q / kv GEMM epilogues
    p       softmax probabilities entering the mix / the PV product           packed in registers inside the cores
    o       attention output (A operand of to_out), GEGLU output (A of FF2)   core / GEGLU epilogues
    u       FF1 output before the gate                                        FF1 epilogue
    y       to_out / FF2 outputs entering the post-LayerNorm                  fast mode stores them in 16 bits

and per block family (s3 = Sparse3DNA block, x = cross-attention block, ff = FeedForward block, lg = final norm + to_logits).
A policy maps (family, class) -> 'f32' | 'bf16' | 'f16'.

    python tools/error_budget.py [--depth 24] [--policies all_bf16,all_f16,...]   ->  one line per policy + a JSON summary
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nuwa_oracle as O  # noqa: E402

FAMS = ('s3', 'x', 'ff', 'lg')
CLASSES = ('h', 'w', 'a', 'p', 'o', 'u', 'y')


def rnd(t, kind):
    if kind == 'f32':
        return t
    if kind == 'bf16':
        return t.to(torch.bfloat16).float()
    if kind == 'f16':
        return t.half().float()
    raise ValueError(kind)


class Policy:
    def __init__(self, default='f32', **over):
        self.m = {(f, c): default for f in FAMS for c in CLASSES}
        for k, v in over.items():
            f, c = k.split('_')
            for ff in (FAMS if f == 'all' else (f,)):
                for cc in (CLASSES if c == 'all' else (c,)):
                    self.m[(ff, cc)] = v

    def __call__(self, t, fam, cls):
        return rnd(t, self.m[(fam, cls)])


def s3_block(h, P, cfg, dil, pol):
    """oracle.sparse3dna (np.py:459-613) with roundings; h = shift(preLN(x)) already rounded"""
    b, n, D = h.shape
    heads = cfg['heads']
    inner = P['to_q.weight'].shape[0]
    d = inner // heads
    idx = O.neighbor_table(cfg['video_shape'], cfg['kernel_size'], dil)
    q = pol(h @ pol(P['to_q.weight'], 's3', 'w').t(), 's3', 'a')
    kv = pol(h @ pol(P['to_kv.weight'], 's3', 'w').t(), 's3', 'a')
    k, v = kv[..., :inner], kv[..., inner:]
    sh = lambda t: t.reshape(b, n, heads, d)
    q, k, v = sh(q), sh(k), sh(v)
    nq = n - 1
    K = idx.shape[1]
    tab = idx[:nq]
    valid = tab >= 0
    gidx = tab.clamp(min=0) + 1
    qs = q[:, 1:] * d ** -0.5
    out = torch.empty(b, nq, heads, d)
    wth = P['talking_heads.weight'].reshape(heads, heads)
    CH = 256                                     # query chunks keep the gathered (nq, J, h, d) tensors small
    for s in range(0, nq, CH):
        e = min(nq, s + CH)
        gi = gidx[s:e].reshape(-1)
        vm = valid[s:e][None, :, :, None, None].float()
        kg = k[:, gi].reshape(b, e - s, K, heads, d) * vm
        vg = v[:, gi].reshape(b, e - s, K, heads, d) * vm
        kk = torch.cat((k[:, :1, None].expand(b, e - s, 1, heads, d), kg), dim=2)
        vv = torch.cat((v[:, :1, None].expand(b, e - s, 1, heads, d), vg), dim=2)
        sim = torch.einsum('bihd,bijhd->bhij', qs[:, s:e], kk)
        mask = F.pad(~valid[s:e], (1, 0), value=False)
        sim = sim.masked_fill(mask[None, None], O.FP32_NEG_MAX)
        attn = sim.softmax(dim=-1, dtype=torch.float32)
        attn = torch.einsum('gh,bhij->bgij', wth, attn)
        attn = pol(attn, 's3', 'p')
        out[:, s:e] = torch.einsum('bgij,bijgd->bigd', attn, vv)
    o = torch.cat((v[:, :1], out), dim=1).reshape(b, n, inner)
    o = pol(o, 's3', 'o')
    return pol(o @ pol(P['to_out.weight'], 's3', 'w').t() + P['to_out.bias'], 's3', 'y')


def x_block(h, P, cfg, ctx, cmask, pol):
    """oracle.attention with context (np.py:315-379) with roundings"""
    b, n, D = h.shape
    heads = cfg['heads']
    inner = P['to_q.weight'].shape[0]
    d = inner // heads
    q = pol(h @ pol(P['to_q.weight'], 'x', 'w').t(), 'x', 'a').reshape(b, n, heads, d)
    kv = pol(pol(ctx, 'x', 'h') @ pol(P['to_kv.weight'], 'x', 'w').t(), 'x', 'a')
    m = ctx.shape[1]
    k = kv[..., :inner].reshape(b, m, heads, d)
    v = kv[..., inner:].reshape(b, m, heads, d)
    nk = pol(P['null_k'].reshape(heads, d), 'x', 'a')[None, None].expand(b, 1, heads, d)
    nv = pol(P['null_v'].reshape(heads, d), 'x', 'a')[None, None].expand(b, 1, heads, d)
    kk, vv = torch.cat((nk, k), 1), torch.cat((nv, v), 1)
    sim = torch.einsum('bihd,bjhd->bhij', q * d ** -0.5, kk)
    km = F.pad(cmask, (1, 0), value=True)
    sim = sim.masked_fill(~km[:, None, None, :], O.FP32_NEG_MAX)
    attn = pol(sim.softmax(dim=-1, dtype=torch.float32), 'x', 'p')       # xattn3/4 pack P before the mix MFMA ...
    attn = torch.einsum('gh,bhij->bgij', P['talking_heads.weight'].reshape(heads, heads), attn)
    attn = pol(attn, 'x', 'p')                                           # ... and P' before the PV MFMA
    o = pol(torch.einsum('bgij,bjgd->bigd', attn, vv).reshape(b, n, inner), 'x', 'o')
    return pol(o @ pol(P['to_out.weight'], 'x', 'w').t(), 'x', 'y')


def ff_block(h, P, pol):
    u = pol(h @ pol(P['net.0.weight'], 'ff', 'w').t(), 'ff', 'u')
    a, g = u.chunk(2, dim=-1)
    gg = pol(a * F.gelu(g), 'ff', 'o')
    return pol(gg @ pol(P['net.3.weight'], 'ff', 'w').t(), 'ff', 'y')


def forward_logits(P, cfg, ids, ctx, cmask, pol):
    x = O.embed_assemble(ids[:, :-1], P, training=False)
    vt = O.sub(P, 'video_transformer')
    fmap = cfg['video_shape'][1]
    for l in range(cfg['depth']):
        L = O.sub(vt, f'layers.{l}')
        dil = cfg['dilations'][l % len(cfg['dilations'])]
        for j, fam in enumerate(('s3', 'x', 'ff')):
            B = O.sub(L, str(j))
            h = O.layer_norm(x, B['prenorm.weight'], B['prenorm.bias'])
            if fam != 'x':
                h = O.shift_video_tokens(h, fmap)
            h = pol(h, fam, 'h')
            if fam == 's3':
                y = s3_block(h, O.sub(B, 'fn.fn'), cfg, dil, pol)
            elif fam == 'x':
                y = x_block(h, O.sub(B, 'fn'), cfg, ctx, cmask, pol)
            else:
                y = ff_block(h, O.sub(B, 'fn.fn'), pol)
            x = x + O.layer_norm(y, B['postnorm.weight'], B['postnorm.bias'])
    hn = pol(O.stable_layer_norm(x, vt['norm.norm.weight'], vt['norm.norm.bias']), 'lg', 'h')
    return hn @ pol(P['to_logits.weight'], 'lg', 'w').t()


POLICIES = {
    'f32': Policy('f32'),
    'all_bf16': Policy('bf16'),
    'all_f16': Policy('f16'),
    # one family in bf16, the rest exact: the per-family budget of the bf16 mode
    'bf16_only_s3': Policy('f32', s3_all='bf16'),
    'bf16_only_x': Policy('f32', x_all='bf16'),
    'bf16_only_ff': Policy('f32', ff_all='bf16'),
    'bf16_only_lg': Policy('f32', lg_all='bf16'),
    # one operand class in bf16, the rest exact
    'bf16_only_h': Policy('f32', all_h='bf16'),
    'bf16_only_w': Policy('f32', all_w='bf16'),
    'bf16_only_a': Policy('f32', all_a='bf16'),
    'bf16_only_p': Policy('f32', all_p='bf16'),
    'bf16_only_o': Policy('f32', all_o='bf16'),
    'bf16_only_u': Policy('f32', all_u='bf16'),
    'bf16_only_y': Policy('f32', all_y='bf16'),
    # fp16 operands with selected exact (= hi+lo) families
    'f16_lg_exact': Policy('f16', lg_all='f32'),
    'f16_y32': Policy('f16', all_y='f32'),
    'f16_y32_lg_exact': Policy('f16', all_y='f32', lg_all='f32'),
    'f16_ff_exact': Policy('f16', ff_all='f32'),
    'f16_attn_exact': Policy('f16', s3_all='f32', x_all='f32'),
    # GEMMs exact (= bf16 hi+lo, 3 MFMAs), attention cores on single fp16 MFMAs: q, k, v and the probabilities rounded to fp16,
    # the core's output leaves as a hi + lo pair (no rounding)
    'x3_gemm_f16_cores': Policy('f32', s3_a='f16', s3_p='f16', x_a='f16', x_p='f16'),
    'x3_gemm_f16_cores_o': Policy('f32', s3_a='f16', s3_p='f16', x_a='f16', x_p='f16', s3_o='f16', x_o='f16'),
    'x3_gemm_bf16_cores': Policy('f32', s3_a='bf16', s3_p='bf16', x_a='bf16', x_p='bf16'),
    # ... and additionally the FeedForward block on single fp16 MFMAs (h, weights, gate output in fp16; its output to the post-norm fp32)
    'f16_cores_f16_ff': Policy('f32', s3_a='f16', s3_p='f16', x_a='f16', x_p='f16', ff_h='f16', ff_w='f16', ff_o='f16'),
    'f16_cores_f16_ff_qkv': Policy('f32', s3_a='f16', s3_p='f16', x_a='f16', x_p='f16', ff_h='f16', ff_w='f16', ff_o='f16', s3_h='f16', s3_w='f16'),
    'f16_ff_only': Policy('f32', ff_h='f16', ff_w='f16', ff_o='f16'),
    # round 4: the 2-MFMA form of the remaining hi + lo products -- fp16 ACTIVATION x exact (fp16 hi + lo) WEIGHT -- on to_out x2 (A operand =
    # the cores' output, class o), the cross-attention q / kv projections (A = LayerNorm output / context, class h) and to_logits (lg_h)
    'cur': Policy('f32', s3_a='f16', s3_p='f16', x_a='f16', x_p='f16', ff_h='f16', ff_w='f16', ff_o='f16', s3_h='f16', s3_w='f16'),
    'cur_o16': Policy('f32', s3_a='f16', s3_p='f16', x_a='f16', x_p='f16', ff_h='f16', ff_w='f16', ff_o='f16', s3_h='f16', s3_w='f16',
                      s3_o='f16', x_o='f16'),
    'cur_o16_xh16': Policy('f32', s3_a='f16', s3_p='f16', x_a='f16', x_p='f16', ff_h='f16', ff_w='f16', ff_o='f16', s3_h='f16', s3_w='f16',
                           s3_o='f16', x_o='f16', x_h='f16'),
    'cur_2mfma_all': Policy('f32', s3_a='f16', s3_p='f16', x_a='f16', x_p='f16', ff_h='f16', ff_w='f16', ff_o='f16', s3_h='f16', s3_w='f16',
                            s3_o='f16', x_o='f16', x_h='f16', lg_h='f16'),
    # y (to_out / FF2 outputs entering the post-LayerNorm) stored as fp16 instead of fp32: 2 bytes per element less in the GEMM epilogue, the
    # LayerNorm forward, the saved activations and the LayerNorm backward
    'cur_y16': Policy('f32', s3_a='f16', s3_p='f16', x_a='f16', x_p='f16', ff_h='f16', ff_w='f16', ff_o='f16', s3_h='f16', s3_w='f16', all_y='f16'),
    'cur_y16_2mfma': Policy('f32', s3_a='f16', s3_p='f16', x_a='f16', x_p='f16', ff_h='f16', ff_w='f16', ff_o='f16', s3_h='f16', s3_w='f16', all_y='f16',
                            s3_o='f16', x_o='f16', x_h='f16', lg_h='f16'),
    # every activation operand fp16, every weight exact: what 2-MFMA products everywhere (FeedForward / qkv included) would give
    'f16_act_exact_w': Policy('f16', all_w='f32', all_y='f32', all_u='f32'),
    # ... and with the FeedForward / qkv weights back in fp16 but u / y exact (= cur_2mfma_all spelled from the other side)
    'f16_act_w16_ffqkv': Policy('f16', all_w='f32', all_y='f32', all_u='f32', ff_w='f16', s3_w='f16'),
    # decomposition of 'cur' by what is rounded (which part of the 6.8e-4 is whose)
    'cur_only_cores': Policy('f32', s3_a='f16', s3_p='f16', x_a='f16', x_p='f16'),
    'cur_only_ff': Policy('f32', ff_h='f16', ff_w='f16', ff_o='f16'),
    'cur_only_ff_w': Policy('f32', ff_w='f16'),
    'cur_only_ff_act': Policy('f32', ff_h='f16', ff_o='f16'),
    'cur_only_qkv': Policy('f32', s3_h='f16', s3_w='f16'),
    'cur_only_qkv_w': Policy('f32', s3_w='f16'),
    # bf16 with the final GEMM exact / the y stores in fp32
    'bf16_y32': Policy('bf16', all_y='f32'),
    'bf16_lg_exact': Policy('bf16', lg_all='f32'),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--depth', type=int, default=24)
    ap.add_argument('--policies', default='all_bf16,all_f16')
    ap.add_argument('--threads', type=int, default=min(os.cpu_count() or 1, 32))
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    import bench
    c = dict(bench.CFGS['cfg3'], dec_depth=args.depth)
    torch.manual_seed(0)
    nuwa = bench.build_model(c, 'cpu')
    P = {k: v.detach().clone() for k, v in nuwa.state_dict().items() if not k.startswith('vae.') and not k.startswith('text_')}
    N = c['frames'] * c['fmap'] ** 2
    g = torch.Generator().manual_seed(11)                  # the inputs of tests/test_gpu_named_size.py::test_cfg3_full_depth_logits_vs_oracle
    ids = torch.randint(0, c['codebook'], (1, N), generator=g)
    ctx = torch.randn(1, c['text_len'], c['dim'], generator=g)
    mask = torch.ones(1, c['text_len'], dtype=torch.bool)
    mask[:, -64:] = torch.rand(1, 64, generator=g) > 0.5
    cfg = dict(video_shape=(c['frames'], c['fmap'], c['fmap']), kernel_size=c['kernel'], dilations=c['dilation'], heads=c['heads'],
               depth=c['dec_depth'], shift=True)
    out = {}
    with torch.no_grad():
        t0 = time.time()
        ref = forward_logits(P, cfg, ids, ctx, mask, POLICIES['f32']).double()
        chk = O.decoder_loss(P, cfg, ids, ctx, mask, training=False, return_logits=True)[1].double()
        print(f'# f32 restatement vs oracle.decoder_loss: rel-max {float((ref - chk).abs().max() / chk.abs().max()):.2e}  ({time.time() - t0:.0f} s)', flush=True)
        for name in args.policies.split(','):
            t0 = time.time()
            got = forward_logits(P, cfg, ids, ctx, mask, POLICIES[name]).double()
            e = float((got - ref).abs().max() / ref.abs().max())
            l2 = float((got - ref).norm() / ref.norm())
            out[name] = dict(logits_rel_max=e, logits_rel_l2=l2)
            print(f'{name:22s} logits rel-max {e:.3e}  rel-l2 {l2:.3e}   ({time.time() - t0:.0f} s)', flush=True)
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(dict(depth=args.depth, policies=out), f, indent=1)


if __name__ == '__main__':
    main()
