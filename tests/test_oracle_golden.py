"""Oracle (oracle/nuwa_oracle.py) against the committed golden fixtures that were captured from
the reference itself (tests/golden/make_golden.py).  Runs anywhere, CPU only."""
import pytest
import torch

from oracle import nuwa_oracle as O
from golden_util import load, load_raw, tup

TOL = dict(rtol=1e-4, atol=2e-5)


def req(P):
    return {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in P.items()}


@pytest.mark.parametrize('ci', range(10))
def test_g1_sparse3dna(ci):
    A, P, G = load(f'g1_sparse3dna_{ci}')
    P = req(P)
    x = A['x'].clone().requires_grad_(True)
    dil = tup(A['dilation']) if A['dilation'].numel() > 1 else int(A['dilation'])
    y = O.sparse3dna(x, P, tup(A['video_shape']), tup(A['kernel_size']), dil, int(A['heads']))
    torch.testing.assert_close(y, A['y'], **TOL)
    y.backward(A['dy'])
    torch.testing.assert_close(x.grad, A['dx'], **TOL)
    for k, g in G.items():
        torch.testing.assert_close(P[k].grad, g, **TOL)


@pytest.mark.parametrize('ci', range(2))
def test_g1b_sparse3dna_rel_pos_bias(ci):
    A, P, G = load(f'g1b_sparse3dna_relpos_{ci}')
    P = req(P)
    x = A['x'].clone().requires_grad_(True)
    y = O.sparse3dna(x, P, tup(A['video_shape']), tup(A['kernel_size']), int(A['dilation']), int(A['heads']))
    torch.testing.assert_close(y, A['y'], **TOL)
    y.backward(A['dy'])
    torch.testing.assert_close(x.grad, A['dx'], **TOL)
    assert any(k.startswith('rel_pos_bias.') for k in G)
    for k, g in G.items():
        torch.testing.assert_close(P[k].grad, g, **TOL)


def test_g2_cross_attention():
    A, P, G = load('g2_cross_attention')
    P = req(P)
    x, ctx = A['x'].clone().requires_grad_(True), A['ctx'].clone().requires_grad_(True)
    y = O.attention(x, P, int(A['heads']), context=ctx, context_mask=A['mask'])
    torch.testing.assert_close(y, A['y'], **TOL)
    y.backward(A['dy'])
    torch.testing.assert_close(x.grad, A['dx'], **TOL)
    torch.testing.assert_close(ctx.grad, A['dctx'], **TOL)
    for k, g in G.items():
        torch.testing.assert_close(P[k].grad, g, **TOL)


@pytest.mark.parametrize('dim', [32, 48])
def test_g3_feedforward(dim):
    A, P, G = load(f'g3_feedforward_{dim}')
    P = req(P)
    x = A['x'].clone().requires_grad_(True)
    y = O.feedforward(x, P)
    torch.testing.assert_close(y, A['y'], **TOL)
    y.backward(A['dy'])
    torch.testing.assert_close(x.grad, A['dx'], **TOL)
    for k, g in G.items():
        torch.testing.assert_close(P[k].grad, g, **TOL)


def test_g4_norms_shift():
    R = load_raw('g4_norms_shift')
    x = R['x'].clone().requires_grad_(True)
    P = {k[3:]: v for k, v in R.items() if k.startswith('sn.')}
    y = O.sandwich(x, P, lambda h: O.shift_video_tokens(h, int(R['fmap'])))
    torch.testing.assert_close(y, R['y'], **TOL)
    y.backward(R['dy'])
    torch.testing.assert_close(x.grad, R['dx'], **TOL)
    x2 = R['x2'].clone().requires_grad_(True)
    y2 = O.stable_layer_norm(x2, R['sl.norm.weight'], R['sl.norm.bias'])
    torch.testing.assert_close(y2, R['y2'], **TOL)
    y2.backward(R['dy2'])
    torch.testing.assert_close(x2.grad, R['dx2'], **TOL)


@pytest.mark.parametrize('name', ['g5_nuwa_tiny', 'g6_nuwa_tiny_reversible'])
def test_g5_g6_nuwa(name):
    A, P, G = load(name)
    P = req(P)
    cfg = dict(video_shape=(3, 4, 4), kernel_size=3, dilations=(1, 2), heads=2, depth=3, shift=True,
               reversible=bool(A['reversible']), text_depth=2, text_heads=2)
    ctx, mask = O.text_encoder(A['text'], P, cfg)
    torch.testing.assert_close(ctx, A['text_embeds'], **TOL)
    loss, logits = O.decoder_loss(P, cfg, A['video_ids'].reshape(2, -1), ctx, mask, return_logits=True)
    torch.testing.assert_close(logits, A['logits'], **TOL)
    torch.testing.assert_close(loss, A['loss'], **TOL)
    loss.backward()
    n = 0
    for k, g in G.items():
        torch.testing.assert_close(P[k].grad, g, rtol=1e-3, atol=2e-5, msg=lambda m, k=k: f'{k}: {m}')
        n += 1
    assert n > 40


VA_CFG = dict(depth=3, heads=2, video_shape=(3, 4, 4), kernel_size=3, dilations=(1, 2), audio_kernel=7, audio_dilations=(1, 2), every=3,
              v_per_frame=16, a_per_frame=4, shift_video=True, shift_audio=True, audio_loss_weight=0.7, text_depth=2, text_heads=2)


@pytest.mark.parametrize('name', ['g9a_video_audio', 'g9b_video_audio_relpos', 'g9c_video_audio_reversible'])
def test_g9_video_audio(name):
    """BASELINE cfg 5 (NUWAVideoAudio, non-reversible dual decoder): oracle vs the reference's loss, both logits and every
    decoder-side gradient"""
    A, P, G = load(name)
    P = req(P)
    b = A['text'].shape[0]
    ctx, mask = O.text_encoder(A['text'], P, VA_CFG)
    torch.testing.assert_close(ctx, A['text_embeds'], **TOL)
    cfg = dict(VA_CFG, reversible=bool(A['reversible'])) if 'reversible' in A else VA_CFG
    loss, vl, al = O.video_audio_loss(P, cfg, A['video_ids'].reshape(b, -1), A['audio_ids'], ctx, mask, return_logits=True)
    torch.testing.assert_close(vl, A['video_logits'], **TOL)
    torch.testing.assert_close(al, A['audio_logits'], **TOL)
    torch.testing.assert_close(loss, A['loss'], **TOL)
    loss.backward()
    n = 0
    for k, g in G.items():
        torch.testing.assert_close(P[k].grad, g, rtol=1e-3, atol=2e-5, msg=lambda m, k=k: f'{k}: {m}')
        n += 1
    assert n > 150


@pytest.mark.parametrize('name', ['g13a_generate_nuwa', 'g13b_generate_nuwa_reversible'])
def test_g13_nuwa_generate(name):
    """the oracle's restatement of NUWA.generate (greedy) samples the token ids the REFERENCE's own generate() produced"""
    A, P, _ = load(name)
    cfg = dict(video_shape=(3, 4, 4), kernel_size=3, dilations=(1, 2), heads=2, depth=3, shift=True, reversible=bool(A['reversible']),
               text_depth=2, text_heads=2)
    with torch.no_grad():
        ids = O.nuwa_generate_greedy(P, cfg, A['text'], A['video_ids'].shape[1], cond_scale=float(A['cond_scale']))
    assert torch.equal(ids, A['video_ids'].long())


@pytest.mark.parametrize('name', ['g13c_generate_video_audio', 'g13d_generate_video_audio_reversible'])
def test_g13_video_audio_generate(name):
    """the same for NUWAVideoAudio.generate: alternating video / audio frames, both dual decoders"""
    A, P, _ = load(name)
    cfg = dict(VA_CFG, reversible=bool(A['reversible']), dilations=(1, 2), audio_dilations=(1, 2))
    with torch.no_grad():
        vids, aids = O.video_audio_generate_greedy(P, cfg, A['text'], 2, cond_scale=float(A['cond_scale']))
    assert torch.equal(vids, A['video_ids'].long()) and torch.equal(aids, A['audio_ids'].long())


SKETCH_CFG = dict(video_shape=(3, 4, 4), sketch_shape=(2, 4, 4), kernel_size=3, dilations=(1, 2), heads=2, enc_heads=2, depth=3,
                  enc_depth=2, shift=True, cross_kernel=3, cross_dilations=(1, 2))


@pytest.mark.parametrize('name', ['g11a_sketch', 'g11b_sketch_reversible_3dna'])
def test_g11_sketch(name):
    """row f4 (NUWASketch): oracle vs the reference's sketch embeddings, logits, loss and every gradient, from the token ids
    the fixture carries (plain stacks / reversible stacks + non-causal 3DNA sketch encoder + masked sketch frame)"""
    A, P, G = load(name)
    P = req(P)
    rev = bool(A['reversible'])
    cfg = dict(SKETCH_CFG, enc_reversible=rev, dec_reversible=rev, enc_3dna=rev)
    smask = A['sketch_mask'] if bool(A['has_mask']) else None
    loss, logits, ctx = O.sketch_loss(P, cfg, A['sketch_ids'], A['video_ids'], smask)
    torch.testing.assert_close(ctx, A['sketch_embeds'], **TOL)
    torch.testing.assert_close(logits, A['logits'], **TOL)
    torch.testing.assert_close(loss, A['loss'], **TOL)
    loss.backward()
    n = 0
    for k, g in G.items():
        torch.testing.assert_close(P[k].grad, g, rtol=1e-3, atol=2e-5, msg=lambda m, k=k: f'{k}: {m}')
        n += 1
    assert n > 40


def test_g13e_sketch_generate():
    """the oracle's restatement of NUWASketch.generate (greedy, guided) samples the token ids the reference's own generate() produced"""
    A, P, _ = load('g13e_generate_sketch')
    cfg = dict(SKETCH_CFG, enc_reversible=False, dec_reversible=False, enc_3dna=False)
    with torch.no_grad():
        ids = O.sketch_generate_greedy(P, cfg, A['sketch_ids'], A['video_ids'].shape[1], cond_scale=float(A['cond_scale']))
    assert torch.equal(ids, A['video_ids'].long())


def test_g10_text_encoder():
    A, P, G = load('g10_text_encoder')
    P = req({k: v for k, v in P.items() if 'net.blocks.' not in k})
    x = A['x'].clone().requires_grad_(True)
    y = O.text_encoder_stack(x, P, 2, 2, A['mask'], A['freqs'])
    torch.testing.assert_close(y, A['y'], **TOL)
    y.backward(A['dy'])
    torch.testing.assert_close(x.grad, A['dx'], rtol=1e-3, atol=2e-5)
    n = 0
    for k, g in G.items():
        if 'net.blocks.' not in k:
            torch.testing.assert_close(P[k].grad, g, rtol=1e-3, atol=2e-5, msg=lambda m, k=k: f'{k}: {m}')
            n += 1
    assert n > 20


def test_g7_vae_encode():
    A, P, _ = load('g7_vae')
    fm = O.vae_encode_fmap(A['img'], P, num_layers=int(A['num_layers']), heads=int(A['heads']))
    torch.testing.assert_close(fm, A['fmap'], **TOL)
    idx, gap = O.vq_eval_lookup(fm, P['vq.embed'], P['vq.project_in.weight'], P['vq.project_in.bias'])
    sure = A['top2_gap'] > 1e-5
    assert torch.equal(idx[sure], A['indices'][sure])   # bit-exact wherever the top-2 gap is not a near-tie
    assert sure.float().mean() > 0.95


def test_g8_decoder_stack():
    A, P, G = load('g8_decoder_stack')
    P = req(P)
    x, ctx = A['x'].clone().requires_grad_(True), A['ctx'].clone().requires_grad_(True)
    cfg = dict(video_shape=(3, 4, 4), kernel_size=3, dilations=(1, 2), heads=2, depth=3, shift=True)
    y = O.decoder_stack(x, P, cfg, ctx, A['mask'])
    torch.testing.assert_close(y, A['y'], **TOL)
    y.backward(A['dy'])
    torch.testing.assert_close(x.grad, A['dx'], rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(ctx.grad, A['dctx'], rtol=1e-3, atol=2e-5)
    for k, g in G.items():
        torch.testing.assert_close(P[k].grad, g, rtol=1e-3, atol=2e-5, msg=lambda m, k=k: f'{k}: {m}')
