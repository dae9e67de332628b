"""Generates tests/golden/*.npz by importing the READ-ONLY reference checkout (through
oracle/ref_shims.py) in the build container and recording seeded inputs, the parameters
(state_dict, reference key names) and the reference's outputs / gradients.

    python tests/golden/make_golden.py [name prefix ...]

The fixtures are DATA (inputs + expected outputs).  They let the GPU box -- where
/root/reference does not exist -- check both the oracle and the HIP path against the
reference's own numbers.  fp32 CPU, dropout 0, cond_dropout_prob 0.
Seeds: torch.manual_seed(0) for parameters, torch.manual_seed(1) for data.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402

ref_shims.install()
from nuwa_pytorch import NUWA, NUWASketch, NUWAVideoAudio, VQGanVAE  # noqa: E402
from nuwa_pytorch.nuwa_pytorch import (Sparse3DNA, Attention, FeedForward, SandwichNorm,  # noqa: E402
                                       ShiftVideoTokens, StableLayerNorm, Transformer, ReversibleTransformer, RotaryEmbedding)


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (np_(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


def params(mod, prefix='p.'):
    return {prefix + k: v for k, v in mod.state_dict().items()}


def grads(mod, prefix='g.'):
    return {prefix + k: p.grad for k, p in mod.named_parameters() if p.grad is not None}


def g1_sparse3dna():
    cases = [((3, 4, 4), 3, 1, None), ((3, 4, 4), 3, 2, None), ((4, 8, 8), (5, 3, 3), 1, None),
             ((4, 8, 8), (5, 3, 3), 2, None), ((4, 8, 8), (5, 3, 3), 4, None),
             ((3, 4, 4), 3, 1, 1), ((3, 4, 4), 3, 1, 2), ((3, 4, 4), 3, 1, 16), ((3, 4, 4), 3, 1, 17),
             ((3, 4, 4), 3, 2, 23)]
    for ci, (shape, kernel, dil, n) in enumerate(cases):
        torch.manual_seed(0)
        m = Sparse3DNA(dim=32, video_shape=shape, kernel_size=kernel, dilation=dil, heads=2, dim_head=32, causal=True)
        N = shape[0] * shape[1] * shape[2]
        n = N if n is None else n
        torch.manual_seed(1)
        x = torch.randn(2, n, 32, requires_grad=True)
        y = m(x)
        g = torch.randn_like(y)
        y.backward(g)
        ks = kernel if isinstance(kernel, tuple) else (kernel,) * 3
        save(f'g1_sparse3dna_{ci}', x=x, y=y, dy=g, dx=x.grad, video_shape=shape, kernel_size=ks, dilation=dil,
             heads=2, **params(m), **grads(m))


def g1b_sparse3dna_rel_pos_bias():
    """Sparse3DNA(rel_pos_bias=True) (np.py:416, 512-516, 542): the axial bias over the key taps, incl. its gradient.
    Batch 1 only: the reference adds an (h, 1, j) bias to a ((b h), i, j) score tensor, which broadcasts for b == 1 alone."""
    for ci, (shape, kernel, dil, n) in enumerate([((4, 8, 8), (5, 3, 3), 2, None), ((3, 4, 4), 3, 1, 23)]):
        torch.manual_seed(0)
        m = Sparse3DNA(dim=32, video_shape=shape, kernel_size=kernel, dilation=dil, heads=2, dim_head=32, causal=True, rel_pos_bias=True)
        N = shape[0] * shape[1] * shape[2]
        n = N if n is None else n
        torch.manual_seed(1)
        x = torch.randn(1, n, 32, requires_grad=True)
        y = m(x)
        g = torch.randn_like(y)
        y.backward(g)
        ks = kernel if isinstance(kernel, tuple) else (kernel,) * 3
        save(f'g1b_sparse3dna_relpos_{ci}', x=x, y=y, dy=g, dx=x.grad, video_shape=shape, kernel_size=ks, dilation=dil,
             heads=2, **params(m), **grads(m))


def g2_cross_attention():
    torch.manual_seed(0)
    m = Attention(dim=32, heads=2, dim_head=32)
    torch.manual_seed(1)
    x = torch.randn(3, 20, 32, requires_grad=True)
    ctx = torch.randn(3, 7, 32, requires_grad=True)
    mask = torch.ones(3, 7, dtype=torch.bool)
    mask[1] = False
    mask[2, 4:] = False
    y = m(x, context=ctx, context_mask=mask)
    g = torch.randn_like(y)
    y.backward(g)
    save('g2_cross_attention', x=x, ctx=ctx, mask=mask, y=y, dy=g, dx=x.grad, dctx=ctx.grad, heads=2,
         **params(m), **grads(m))


def g3_feedforward():
    for dim in (32, 48):
        torch.manual_seed(0)
        m = FeedForward(dim=dim)
        torch.manual_seed(1)
        x = torch.randn(2, 9, dim, requires_grad=True)
        y = m(x)
        g = torch.randn_like(y)
        y.backward(g)
        save(f'g3_feedforward_{dim}', x=x, y=y, dy=g, dx=x.grad, **params(m), **grads(m))


def g4_norms_and_shift():
    torch.manual_seed(0)
    sn = SandwichNorm(dim=32, fn=ShiftVideoTokens(torch.nn.Identity(), image_size=4))
    sl = StableLayerNorm(32)
    with torch.no_grad():
        for p in list(sn.parameters()) + list(sl.parameters()):
            p.normal_()
    torch.manual_seed(1)
    x = torch.randn(2, 23, 32, requires_grad=True)
    y = sn(x)
    g = torch.randn_like(y)
    y.backward(g)
    x2 = torch.randn(2, 9, 32, requires_grad=True)
    y2 = sl(x2)
    g2 = torch.randn_like(y2)
    y2.backward(g2)
    save('g4_norms_shift', x=x, y=y, dy=g, dx=x.grad, fmap=4, x2=x2, y2=y2, dy2=g2, dx2=x2.grad,
         **params(sn, 'sn.'), **grads(sn, 'gsn.'), **params(sl, 'sl.'), **grads(sl, 'gsl.'))


def tiny_nuwa(reversible):
    torch.manual_seed(0)
    vae = VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32,
                   use_vgg_and_gan=False)
    return NUWA(vae=vae, dim=32, text_num_tokens=50, text_max_seq_len=8, max_video_frames=3, text_enc_depth=2,
                dec_depth=3, enc_reversible=True, dec_reversible=reversible, dec_heads=2, dec_dim_head=32,
                text_enc_heads=2, text_enc_dim_head=16, sparse_3dna_kernel_size=3, sparse_3dna_dilation=(1, 2))


def g5_g6_nuwa():
    for name, rev in (('g5_nuwa_tiny', False), ('g6_nuwa_tiny_reversible', True)):
        nuwa = tiny_nuwa(rev)
        torch.manual_seed(1)
        text = torch.randint(1, 50, (2, 8))
        text[1, 5:] = 0
        vid = torch.randint(0, 64, (2, 3, 4, 4))
        cap = {}
        hk = nuwa.to_logits.register_forward_hook(lambda m, i, o: cap.__setitem__('logits', o.detach()))
        hk2 = nuwa.text_transformer.register_forward_hook(lambda m, i, o: cap.__setitem__('ctx', o.detach()))
        loss = nuwa(text=text, video=vid, return_loss=True, cond_dropout_prob=0.)
        hk.remove(); hk2.remove()
        loss.backward()
        P = {k: v for k, v in params(nuwa).items() if not k.startswith('p.vae.') and '.net.blocks.' not in k}
        G = {k: v for k, v in grads(nuwa).items() if '.net.blocks.' not in k}
        save(name, text=text, video_ids=vid, loss=loss, logits=cap['logits'], text_embeds=cap['ctx'],
             reversible=rev, **P, **G)


def g7_vae():
    torch.manual_seed(0)
    vae = VQGanVAE(dim=32, image_size=32, num_layers=2, vq_codebook_size=64, vq_codebook_dim=16,
                   use_vgg_and_gan=False, attn_dim_head=16, attn_heads=4).eval()
    torch.manual_seed(1)
    img = torch.rand(3, 3, 32, 32)
    fm = img
    stages = {}
    with torch.no_grad():
        for i, enc in enumerate(vae.encoders):
            fm = enc(fm)
            stages[f'stage{i}'] = fm
        quant, ind, _ = vae.vq(fm)          # VQ = shimmed restatement: PARITY UNPINNED
        xn = torch.nn.functional.normalize(vae.vq.project_in(fm.permute(0, 2, 3, 1)), dim=-1)
        sim = xn @ torch.nn.functional.normalize(vae.vq.embed, dim=-1).t()
        top2 = sim.topk(2, dim=-1).values
        recon = vae.decode(quant)
        loss = vae(img, return_loss=True)
    save('g7_vae', img=img, fmap=fm, indices=ind, top2_gap=top2[..., 0] - top2[..., 1], recon=recon,
         recon_loss=loss, num_layers=2, heads=4, **stages, **params(vae))


def g12_vae_cfg1():
    """BASELINE cfg 1 exactly: VQGanVAE dim=64 image_size=32 num_layers=2, batch 4 random images, forward(return_loss=True) on CPU.
    Weights come from tests/golden_util.fill_params (name-seeded, so the fixture carries no parameters).  Two cases:
    eval (frozen codebook, no straight-through: gradients reach the decoder only) and train with vq_kmeans_init=False (EMA codebook
    update + straight-through, gradients reach the encoder too; this leg runs through the restated VectorQuantize: PARITY UNPINNED)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from golden_util import fill_params, sample2048
    out = {}
    for mode in ('eval', 'train'):
        vae = VQGanVAE(dim=64, image_size=32, num_layers=2, use_vgg_and_gan=False, vq_kmeans_init=False)
        fill_params(vae, seed=12)
        vae.train(mode == 'train')
        torch.manual_seed(1)
        img = torch.rand(4, 3, 32, 32)
        loss, recon = vae(img, return_loss=True, return_recons=True)
        loss.backward()
        G = {k: p.grad for k, p in vae.named_parameters() if p.grad is not None}
        out.update({f'{mode}.loss': loss, f'{mode}.recon': recon, 'img': img})
        for k in ('decoders.4.weight', 'decoders.0.net.0.weight', 'decoders.1.to_qkv.weight', 'vq.project_out.weight',
                  'encoders.0.weight', 'encoders.2.0.bias', 'encoders.4.post_norm.g', 'vq.project_in.weight'):
            if k in G:                                    # a strided sample + the norm keep the fixture small
                out[f'{mode}.g.{k}'] = sample2048(G[k])
                out[f'{mode}.gnorm.{k}'] = G[k].norm()
        out[f'{mode}.n_grads'] = len(G)
        if mode == 'train':
            out['train.embed_after'] = sample2048(vae.vq.embed)
            out['train.cluster_size_after'] = vae.vq.cluster_size
    save('g12_vae_cfg1', **out)


def g8_decoder_layer():
    torch.manual_seed(0)
    tr = Transformer(dim=32, depth=3, causal=True, heads=2, dim_head=32, cross_attend=True,
                     sparse_3dna_attn=True, sparse_3dna_kernel_size=(3, 3, 3), sparse_3dna_video_shape=(3, 4, 4),
                     sparse_3dna_dilations=(1, 2), shift_video_tokens=True)
    torch.manual_seed(1)
    x = torch.randn(2, 48, 32, requires_grad=True)
    ctx = torch.randn(2, 6, 32, requires_grad=True)
    mask = torch.ones(2, 6, dtype=torch.bool)
    mask[1, 3:] = False
    y = tr(x, context=ctx, context_mask=mask)
    g = torch.randn_like(y)
    y.backward(g)
    save('g8_decoder_stack', x=x, ctx=ctx, mask=mask, y=y, dy=g, dx=x.grad, dctx=ctx.grad,
         **params(tr), **grads(tr))


def g10_text_encoder():
    """row f1: the text encoder's blocks -- non-causal self-attention (null k/v, key mask, talking heads, rotary on q, k AND v)
    + FeedForward in a ReversibleTransformer, at a head size the HIP attention kernels cover (32)"""
    torch.manual_seed(0)
    tr = ReversibleTransformer(dim=32, depth=2, heads=2, dim_head=32)
    rot = RotaryEmbedding(dim=32)
    torch.manual_seed(1)
    x = torch.randn(3, 12, 32, requires_grad=True)
    mask = torch.ones(3, 12, dtype=torch.bool)
    mask[1, 7:] = False
    mask[2, 1:] = False
    freqs = rot(12, device=x.device)
    y = tr(x, mask=mask, rotary_pos_emb=freqs)
    g = torch.randn_like(y)
    y.backward(g)
    P = {k: v for k, v in params(tr).items() if '.net.blocks.' not in k}
    G = {k: v for k, v in grads(tr).items() if '.net.blocks.' not in k}
    save('g10_text_encoder', x=x, mask=mask, freqs=freqs, y=y, dy=g, dx=x.grad, **P, **G)


VA_KW = dict(dim=32, image_size=16, num_audio_tokens=40, num_audio_tokens_per_video_frame=4, max_video_frames=3, text_num_tokens=50,
             text_max_seq_len=8, text_enc_depth=2, text_enc_dim_head=16, text_enc_heads=2, enc_reversible=True, dec_reversible=False,
             dec_depth=3, dec_dim_head=32, dec_heads=2, sparse_3dna_kernel_size=3, sparse_3dna_dilation=2, sparse_2dna_kernel_size=7,
             sparse_2dna_dilation=2, cross_modality_attn_every=3, audio_loss_weight=0.7)


def g9_video_audio(only=None):
    """BASELINE cfg 5 path, tiny: NUWAVideoAudio.forward(return_loss=True) with the non-reversible DualModalityDecoder.
    a: no 3DNA rel-pos bias, batch 2 (padded text);  b: the default 3DNA rel-pos bias, batch 1 (the reference's bias add only
    broadcasts for one sample)."""
    for name, rel, b, rev in (('g9a_video_audio', False, 2, False), ('g9b_video_audio_relpos', True, 1, False),
                              ('g9c_video_audio_reversible', False, 2, True)):
        if only is not None and name not in only:
            continue
        torch.manual_seed(0)
        vae = VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
        m = NUWAVideoAudio(vae=vae, sparse_3dna_rel_pos_bias=rel, **{**VA_KW, 'dec_reversible': rev})
        torch.manual_seed(1)
        text = torch.randint(1, 50, (b, 8))
        text[-1, 5:] = 0
        vid = torch.randint(0, 64, (b, 3, 4, 4))
        aud = torch.randint(0, 40, (b, 12))
        cap = {}
        hooks = [m.to_video_logits.register_forward_hook(lambda mod, i, o: cap.__setitem__('vl', o.detach())),
                 m.to_audio_logits.register_forward_hook(lambda mod, i, o: cap.__setitem__('al', o.detach())),
                 m.text_transformer.register_forward_hook(lambda mod, i, o: cap.__setitem__('ctx', o.detach()))]
        loss = m(text=text, video=vid, audio=aud, return_loss=True, cond_dropout_prob=0.)
        for h in hooks:
            h.remove()
        loss.backward()
        P = {k: v for k, v in params(m).items() if not k.startswith('p.vae.') and '.net.blocks.' not in k}
        G = {k: v for k, v in grads(m).items() if '.net.blocks.' not in k}
        save(name, text=text, video_ids=vid, audio_ids=aud, loss=loss, video_logits=cap['vl'], audio_logits=cap['al'],
             text_embeds=cap['ctx'], rel_pos_bias=rel, reversible=rev, **P, **G)



def _codes_to_ids(codes, codebook):
    """rows of the codebook back to their indices (exact match): codes [n, d, h, w] as the reference hands them to vae.decode"""
    flat = codes.permute(0, 2, 3, 1).reshape(-1, codes.shape[1])
    d = (flat[:, None, :] - codebook[None]).abs().amax(-1)
    ids = d.argmin(-1)
    assert float(d.gather(1, ids[:, None]).max()) == 0.
    return ids


def g13_generate():
    """The reference's OWN generate() (np.py:1841-1915 and 2111-2222) under greedy sampling (filter_thres 0.99 keeps one logit, so
    the Gumbel noise cannot change the arg-max): the sampled video (and audio) token ids for tiny models with recorded parameters.
    The ids handed to vae.decode are recovered from its input (rows of the codebook)."""
    for name, kind, rev, cs in (('g13a_generate_nuwa', 'nuwa', False, 2.), ('g13b_generate_nuwa_reversible', 'nuwa', True, 2.),
                                ('g13c_generate_video_audio', 'va', False, 2.), ('g13d_generate_video_audio_reversible', 'va', True, 1.)):
        torch.manual_seed(0)
        if kind == 'nuwa':
            m = tiny_nuwa(rev)
        else:
            vae = VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
            m = NUWAVideoAudio(vae=vae, sparse_3dna_rel_pos_bias=False, **{**VA_KW, 'dec_reversible': rev})
        m.eval()
        torch.manual_seed(1)
        text = torch.randint(1, 50, (2, 8))
        text[-1, 5:] = 0
        seen = []
        orig = m.vae.decode
        # (only the ids matter here; the tiny VAE's codebook dim differs from its decoder width, which the reference's decode() rejects)
        m.vae.decode = lambda codes: (seen.append(codes.detach().clone()), torch.zeros(codes.shape[0], 3, 16, 16))[1]
        torch.manual_seed(2)
        out = m.generate(text=text, filter_thres=0.99, cond_scale=cs, num_frames=2)
        m.vae.decode = orig
        ids = _codes_to_ids(torch.cat(seen, 0), m.vae.codebook).reshape(2, -1)
        P = {k: v for k, v in params(m).items() if not k.startswith('p.vae.') and '.net.blocks.' not in k}
        extra = dict(audio_ids=out[1]) if kind == 'va' else {}
        save(name, text=text, video_ids=ids, cond_scale=cs, reversible=rev, **extra, **P)


SKETCH_KW = dict(dim=32, image_size=16, max_video_frames=3, sketch_max_video_frames=2, sketch_enc_depth=2, sketch_enc_dim_head=16,
                 sketch_enc_heads=2, dec_depth=3, dec_dim_head=32, dec_heads=2, cross_2dna_kernel_size=3, cross_2dna_dilation=2,
                 sparse_3dna_kernel_size=3, sparse_3dna_dilation=(1, 2))


def g11_sketch():
    """NUWASketch.forward(return_loss=True), tiny (row f4).  a: plain stacks, no sketch mask;  b: reversible encoder and decoder,
    non-causal Sparse3DNA sketch encoder, second sketch frame of sample 1 masked.  The token ids the reference's (randomly
    initialised) VAEs produced are recorded so that the decoder parity does not hinge on the tokenizer."""
    for name, extra, use_mask in (('g11a_sketch', {}, False),
                                  ('g11b_sketch_reversible_3dna', dict(enc_reversible=True, dec_reversible=True,
                                                                       sketch_enc_use_sparse_3dna=True), True)):
        torch.manual_seed(0)
        vae = VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
        sketch_vae = VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=48, vq_codebook_dim=32, use_vgg_and_gan=False)
        m = NUWASketch(vae=vae, sketch_vae=sketch_vae, **{**SKETCH_KW, **extra})
        torch.manual_seed(1)
        sketch = torch.rand(2, 2, 3, 16, 16)
        video = torch.rand(2, 3, 3, 16, 16)
        smask = None
        if use_mask:
            smask = torch.ones(2, 2, dtype=torch.bool)
            smask[1, 1] = False
        cap = {}
        hooks = [m.to_logits.register_forward_hook(lambda mod, i, o: cap.__setitem__('logits', o.detach())),
                 m.sketch_transformer.register_forward_hook(lambda mod, i, o: cap.__setitem__('ctx', o.detach()))]
        with torch.no_grad():
            sketch_ids = m.sketch_vae.get_video_indices(sketch)
            video_ids = m.vae.get_video_indices(video)
        loss = m(sketch=sketch, sketch_mask=smask, video=video, return_loss=True, cond_dropout_prob=0.)
        for h in hooks:
            h.remove()
        loss.backward()
        P = {k: v for k, v in params(m).items() if '.net.blocks.' not in k and not k.startswith(('p.vae.', 'p.sketch_vae.'))}
        G = {k: v for k, v in grads(m).items() if '.net.blocks.' not in k}
        save(name, sketch_ids=sketch_ids, video_ids=video_ids, loss=loss, logits=cap['logits'], sketch_embeds=cap['ctx'],
             sketch_mask=(smask if use_mask else torch.ones(2, 2, dtype=torch.bool)), has_mask=use_mask,
             reversible=bool(extra), **P, **G)


def g13e_generate_sketch():
    """NUWASketch.generate (np.py:2438-2511), greedy, guided: the sampled token ids; the sketch token ids of the (randomly initialised)
    sketch VAE are recorded so that the product side can be fed the same context"""
    torch.manual_seed(0)
    vae = VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
    sketch_vae = VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=48, vq_codebook_dim=32, use_vgg_and_gan=False)
    m = NUWASketch(vae=vae, sketch_vae=sketch_vae, **SKETCH_KW).eval()
    torch.manual_seed(1)
    sketch = torch.rand(2, 2, 3, 16, 16)
    with torch.no_grad():
        sketch_ids = m.sketch_vae.get_video_indices(sketch)
    seen = []
    m.vae.decode = lambda codes: (seen.append(codes.detach().clone()), torch.zeros(codes.shape[0], 3, 16, 16))[1]
    torch.manual_seed(2)
    m.generate(sketch=sketch, filter_thres=0.99, cond_scale=2., num_frames=2)
    ids = _codes_to_ids(torch.cat(seen, 0), m.vae.codebook).reshape(2, -1)
    P = {k: v for k, v in params(m).items() if '.net.blocks.' not in k and not k.startswith(('p.vae.', 'p.sketch_vae.'))}
    save('g13e_generate_sketch', sketch=sketch, sketch_ids=sketch_ids, video_ids=ids, cond_scale=2., **P)


if __name__ == '__main__':
    makers = [g1_sparse3dna, g1b_sparse3dna_rel_pos_bias, g2_cross_attention, g3_feedforward, g4_norms_and_shift, g5_g6_nuwa, g7_vae,
              g8_decoder_layer, g9_video_audio, g10_text_encoder, g11_sketch, g12_vae_cfg1, g13_generate, g13e_generate_sketch]
    want = sys.argv[1:]                       # e.g. `python tests/golden/make_golden.py g12` regenerates only the g12 fixture
    for fn in makers:
        if not want or any(fn.__name__.startswith(w) for w in want):
            fn()
