"""Module-level parity on the MI355X against the golden fixtures captured from the reference
(tests/golden/*.npz): the product modules of nuwa_pytorch_amd, loaded with the reference's own
state dict, must reproduce the reference's outputs and gradients.

Tolerances (max-abs error / max-abs reference):
  'bf16x3'     parity mode    : 1e-3  -- the north-star bound on logits ("within 1e-3 relative"); grads 2e-3
  'bf16x3-fwd' compliant mode : outputs 1e-3 (the bf16x3 forward), gradients as 'bf16' (bf16 backward on the hi parts)
  'bf16'       fast mode      : outputs 2e-2, gradients 7e-2 = 1.5 x the largest errors these tiny fixtures measured in round 2
                                (1.34e-2 on g9a's audio logits; 4.46e-2 on one 2x2 talking-heads gradient of g8, everything else
                                <= 3.3e-2; the reference's own bf16 autocast is 4e-3..1e-2 per layer)
VQ code indices: bit-exact wherever the fixture's top-2 similarity gap exceeds 1e-5."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load, load_raw, tup  # noqa: E402
from gpu_util import report  # noqa: E402

DEV = 'cuda'
MODES = [('bf16x3', 1e-3, 2e-3), ('bf16x3-fwd', 1e-3, 7e-2), ('bf16', 2e-2, 7e-2)]


@pytest.fixture(scope='module')
def A():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import nuwa_pytorch_amd
    return nuwa_pytorch_amd


def run_mode(A, mode):
    A.set_precision(mode)


def check_grads(mod, G, tol, tag, skip=()):
    named = dict(mod.named_parameters())
    n = 0
    for k, g in G.items():
        if any(s in k for s in skip):
            continue
        p = named[k]
        assert p.grad is not None, f'{tag}: no grad for {k}'
        report(f'{tag}.grad.{k}', p.grad, g, tol)
        n += 1
    return n


@pytest.mark.parametrize('mode,tol,gtol', MODES)
@pytest.mark.parametrize('ci', range(10))
def test_g1_sparse3dna_module(A, ci, mode, tol, gtol):
    Ar, P, G = load(f'g1_sparse3dna_{ci}')
    dil = tup(Ar['dilation']) if Ar['dilation'].numel() > 1 else int(Ar['dilation'])
    m = A.Sparse3DNA(dim=32, video_shape=tup(Ar['video_shape']), kernel_size=tup(Ar['kernel_size']), dilation=dil,
                     heads=int(Ar['heads']), dim_head=32, causal=True)
    m.load_state_dict(P)
    m = m.to(DEV)
    run_mode(A, mode)
    try:
        x = Ar['x'].to(DEV).requires_grad_(True)
        y = m(x)
        report(f'g1[{ci},{mode}].y', y, Ar['y'], tol)
        y.backward(Ar['dy'].to(DEV))
        report(f'g1[{ci},{mode}].dx', x.grad, Ar['dx'], gtol)
        check_grads(m, G, gtol, f'g1[{ci},{mode}]')
    finally:
        A.set_precision('bf16')


@pytest.mark.parametrize('mode,tol,gtol', MODES)
def test_g2_cross_attention_module(A, mode, tol, gtol):
    Ar, P, G = load('g2_cross_attention')
    m = A.Attention(dim=32, heads=int(Ar['heads']), dim_head=32)
    m.load_state_dict(P)
    m = m.to(DEV)
    run_mode(A, mode)
    try:
        x, ctx = Ar['x'].to(DEV).requires_grad_(True), Ar['ctx'].to(DEV).requires_grad_(True)
        y = m(x, context=ctx, context_mask=Ar['mask'].to(DEV))
        report(f'g2[{mode}].y', y, Ar['y'], tol)
        y.backward(Ar['dy'].to(DEV))
        report(f'g2[{mode}].dx', x.grad, Ar['dx'], gtol)
        report(f'g2[{mode}].dctx', ctx.grad, Ar['dctx'], gtol)
        check_grads(m, G, gtol, f'g2[{mode}]')
    finally:
        A.set_precision('bf16')


@pytest.mark.parametrize('mode,tol,gtol', MODES)
@pytest.mark.parametrize('dim', [32, 48])
def test_g3_feedforward_module(A, dim, mode, tol, gtol):
    Ar, P, G = load(f'g3_feedforward_{dim}')
    m = A.FeedForward(dim=dim)
    m.load_state_dict(P)
    m = m.to(DEV)
    run_mode(A, mode)
    try:
        x = Ar['x'].to(DEV).requires_grad_(True)
        y = m(x)
        report(f'g3[{dim},{mode}].y', y, Ar['y'], tol)
        y.backward(Ar['dy'].to(DEV))
        report(f'g3[{dim},{mode}].dx', x.grad, Ar['dx'], gtol)
        check_grads(m, G, gtol, f'g3[{dim},{mode}]')
    finally:
        A.set_precision('bf16')


@pytest.mark.parametrize('mode,tol,gtol', MODES)
@pytest.mark.parametrize('dim', [64, 512])       # (512: the fp16-operand forward of 'bf16x3-fwd' takes the product; 64: the generic ring)
def test_feedforward_dropout_runs_on_the_library(A, monkeypatch, dim, mode, tol, gtol):
    """FeedForward with ff_dropout > 0 in training (np.py:260-286: Linear -> GEGLU -> nn.Dropout -> Linear): both products, the gate and its backward
    on libamdnuwa, the keep mask kept for the backward.  Checked against the same arithmetic in torch fp32 with the SAME mask (the mask itself comes
    from torch's RNG stream in production; here ops._ff_keep_mask is replaced by a fixed one), forward and every gradient; in eval mode the module
    must give the dropout-free result."""
    from nuwa_pytorch_amd import ops
    import torch.nn.functional as F
    torch.manual_seed(11)
    p = 0.25
    m = A.FeedForward(dim=dim, dropout=p).to(DEV).train()
    w1, w2 = m.net[0].weight, m.net[3].weight
    FFI = w2.shape[1]
    x = torch.randn(2, 48, dim, device=DEV, requires_grad=True)
    dy = torch.randn(2, 48, dim, device=DEV)
    keep_all = torch.rand(96, (FFI + 31) // 32 * 32, device=DEV) >= p
    monkeypatch.setattr(ops, '_ff_keep_mask', lambda R, C, pp, device: keep_all[:R, :C].clone())
    run_mode(A, mode)
    try:
        y = m(x)
        y.backward(dy)
        got = (y.detach().clone(), x.grad.clone(), w1.grad.clone(), w2.grad.clone())
        for t in (x, w1, w2):
            t.grad = None
        u = x.reshape(96, dim) @ w1.t()
        a, g = u.chunk(2, dim=-1)
        gg = a * F.gelu(g)
        gg = torch.where(keep_all[:, :FFI], gg * (1.0 / (1.0 - p)), torch.zeros((), device=DEV))
        yr = (gg @ w2.t()).reshape(2, 48, dim)
        yr.backward(dy)
        report(f'ff_dropout[{dim},{mode}].y', got[0], yr.detach(), tol)
        report(f'ff_dropout[{dim},{mode}].dx', got[1], x.grad, gtol)
        report(f'ff_dropout[{dim},{mode}].dw1', got[2], w1.grad, gtol)
        report(f'ff_dropout[{dim},{mode}].dw2', got[3], w2.grad, gtol)
        assert float((got[0] - m.eval()(x).detach()).abs().max()) > 1e-3          # (the mask did something)
        m.eval()
        ye = m(x).detach()
        yn = (torch.nn.functional.linear(a * F.gelu(g), w2)).reshape(2, 48, dim).detach()
        report(f'ff_dropout[{dim},{mode}].eval', ye, yn, tol)
    finally:
        A.set_precision('bf16')


def test_g4_sandwich_shift_and_stable_ln(A):
    R = load_raw('g4_norms_shift')
    sn = A.SandwichNorm(dim=32, fn=A.ShiftVideoTokens(torch.nn.Identity(), image_size=int(R['fmap'])))
    sn.load_state_dict({k[3:]: v for k, v in R.items() if k.startswith('sn.')})
    sn = sn.to(DEV)
    x = R['x'].to(DEV).requires_grad_(True)
    y = sn(x)
    report('g4.sandwich.y', y, R['y'], 1e-5)
    y.backward(R['dy'].to(DEV))
    report('g4.sandwich.dx', x.grad, R['dx'], 2e-5)
    sl = A.StableLayerNorm(32)
    sl.load_state_dict({k[3:]: v for k, v in R.items() if k.startswith('sl.')})
    sl = sl.to(DEV)
    A.set_precision('bf16x3')
    try:
        x2 = R['x2'].to(DEV).requires_grad_(True)
        y2 = sl(x2)
        report('g4.stable_ln.y', y2, R['y2'], 3e-5)
        y2.backward(R['dy2'].to(DEV))
        report('g4.stable_ln.dx', x2.grad, R['dx2'], 3e-5)
    finally:
        A.set_precision('bf16')


@pytest.mark.parametrize('mode,tol,gtol', MODES)
def test_g8_decoder_stack(A, mode, tol, gtol):
    Ar, P, G = load('g8_decoder_stack')
    tr = A.Transformer(dim=32, depth=3, causal=True, heads=2, dim_head=32, cross_attend=True, sparse_3dna_attn=True,
                       sparse_3dna_kernel_size=(3, 3, 3), sparse_3dna_video_shape=(3, 4, 4), sparse_3dna_dilations=(1, 2),
                       shift_video_tokens=True)
    tr.load_state_dict(P)
    tr = tr.to(DEV)
    run_mode(A, mode)
    try:
        x, ctx = Ar['x'].to(DEV).requires_grad_(True), Ar['ctx'].to(DEV).requires_grad_(True)
        y = tr(x, context=ctx, context_mask=Ar['mask'].to(DEV))
        report(f'g8[{mode}].y', y, Ar['y'], tol)
        y.backward(Ar['dy'].to(DEV))
        report(f'g8[{mode}].dx', x.grad, Ar['dx'], gtol)
        report(f'g8[{mode}].dctx', ctx.grad, Ar['dctx'], gtol)
        check_grads(tr, G, gtol, f'g8[{mode}]')
    finally:
        A.set_precision('bf16')


@pytest.mark.parametrize('mode,tol,gtol', MODES)
@pytest.mark.parametrize('ci', range(2))
def test_g1b_sparse3dna_rel_pos_bias(A, ci, mode, tol, gtol):
    """Sparse3DNA(rel_pos_bias=True): the axial per-tap, per-head bias and its gradient against the reference fixture"""
    Ar, P, G = load(f'g1b_sparse3dna_relpos_{ci}')
    m = A.Sparse3DNA(dim=32, video_shape=tup(Ar['video_shape']), kernel_size=tup(Ar['kernel_size']), dilation=int(Ar['dilation']),
                     heads=int(Ar['heads']), dim_head=32, causal=True, rel_pos_bias=True)
    m.load_state_dict(P)
    m = m.to(DEV)
    run_mode(A, mode)
    try:
        x = Ar['x'].to(DEV).requires_grad_(True)
        y = m(x)
        report(f'g1b[{ci},{mode}].y', y, Ar['y'], tol)
        y.backward(Ar['dy'].to(DEV))
        report(f'g1b[{ci},{mode}].dx', x.grad, Ar['dx'], gtol)
        assert check_grads(m, G, gtol, f'g1b[{ci},{mode}]') >= 8
    finally:
        A.set_precision('bf16')


def _tiny_nuwa(A, reversible):
    vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
    return A.NUWA(vae=vae, dim=32, text_num_tokens=50, text_max_seq_len=8, max_video_frames=3, text_enc_depth=2,
                  dec_depth=3, enc_reversible=True, dec_reversible=reversible, dec_heads=2, dec_dim_head=32,
                  text_enc_heads=2, text_enc_dim_head=16, sparse_3dna_kernel_size=3, sparse_3dna_dilation=(1, 2))


@pytest.mark.parametrize('mode,tol,gtol', MODES)
@pytest.mark.parametrize('name', ['g5_nuwa_tiny', 'g6_nuwa_tiny_reversible'])
def test_g5_g6_nuwa_loss_logits_grads(A, name, mode, tol, gtol):
    Ar, P, G = load(name)
    nuwa = _tiny_nuwa(A, bool(Ar['reversible']))
    missing, unexpected = nuwa.load_state_dict(P, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith('vae.') or '.net.blocks.' in k for k in missing), missing
    nuwa = nuwa.to(DEV).train()
    run_mode(A, mode)
    try:
        text, vid = Ar['text'].to(DEV), Ar['video_ids'].to(DEV)
        logits = nuwa(text=text, video=vid.reshape(2, -1)[:, :-1], return_loss=False, cond_dropout_prob=0.)
        report(f'{name}[{mode}].logits', logits, Ar['logits'], tol)
        loss = nuwa(text=text, video=vid, return_loss=True, cond_dropout_prob=0.)
        report(f'{name}[{mode}].loss', loss.reshape(1), Ar['loss'].reshape(1), tol)
        loss.backward()
        n = check_grads(nuwa, G, gtol * 2, f'{name}[{mode}]', skip=('.net.blocks.',))
        assert n > 40
    finally:
        A.set_precision('bf16')


def test_g5_with_the_fused_hi_lo_cross_entropy(A, monkeypatch):
    """'bf16x3-fwd' with the opt-in fused to_logits + cross entropy (AMDNUWA_FUSE_LINEAR_CE_X3: no fp32 logits in memory, dlogits from
    one fp16 MFMA per product): the reference's loss and gradients of fixture g5 inside the mode's usual bounds, and the loss equal to
    the default (unfused) path's to 2e-6"""
    from nuwa_pytorch_amd import ops, kernels as K
    Ar, P, G = load('g5_nuwa_tiny')
    nuwa = _tiny_nuwa(A, False)
    nuwa.load_state_dict(P, strict=False)
    nuwa = nuwa.to(DEV).train()
    run_mode(A, 'bf16x3-fwd')
    try:
        text, vid = Ar['text'].to(DEV), Ar['video_ids'].to(DEV)
        base = nuwa(text=text, video=vid, return_loss=True, cond_dropout_prob=0.).detach()
        monkeypatch.setattr(ops, 'FUSE_LINEAR_CE_X3', True)
        seen = []
        real = K.linear_ce
        monkeypatch.setattr(K, 'linear_ce', lambda *a, **k: (seen.append(k.get('w16') is not None), real(*a, **k))[1])
        loss = nuwa(text=text, video=vid, return_loss=True, cond_dropout_prob=0.)
        assert seen == [True]                               # the fused kernels took the call, with the fp16 dlogits pass
        report('g5[bf16x3-fwd, fused ce].loss', loss.reshape(1), Ar['loss'].reshape(1), 1e-3)
        report('g5[bf16x3-fwd, fused ce].loss_vs_unfused', loss.detach().reshape(1), base.reshape(1), 2e-6)
        loss.backward()
        assert check_grads(nuwa, G, 7e-2 * 2, 'g5[bf16x3-fwd, fused ce]', skip=('.net.blocks.',)) > 40
    finally:
        A.set_precision('bf16')


def test_fused_cross_entropy_auto_policy_follows_the_memory_headroom(A, monkeypatch):
    """AMDNUWA_FUSE_LINEAR_CE_X3 = auto (opt-in): the fused form exactly when the fp32 logits + dlogits would push the device past
    the policy's fraction of its memory (what keeps the whole b = 128 step out of allocator retries), the unfused one otherwise"""
    from nuwa_pytorch_amd import ops, kernels as K
    Ar, P, G = load('g5_nuwa_tiny')
    nuwa = _tiny_nuwa(A, False)
    nuwa.load_state_dict(P, strict=False)
    nuwa = nuwa.to(DEV).train()
    run_mode(A, 'bf16x3-fwd')
    try:
        text, vid = Ar['text'].to(DEV), Ar['video_ids'].to(DEV)
        monkeypatch.setattr(ops, 'FUSE_LINEAR_CE_X3', 'auto')
        seen = []
        real = K.linear_ce
        monkeypatch.setattr(K, 'linear_ce', lambda *a, **k: (seen.append(1), real(*a, **k))[1])
        monkeypatch.setattr(ops, 'FUSE_LINEAR_CE_X3_MEM_FRAC', 10.0)           # never tight
        a = nuwa(text=text, video=vid, return_loss=True, cond_dropout_prob=0.).detach()
        assert seen == []
        monkeypatch.setattr(ops, 'FUSE_LINEAR_CE_X3_MEM_FRAC', 0.0)            # always tight
        b = nuwa(text=text, video=vid, return_loss=True, cond_dropout_prob=0.).detach()
        assert seen == [1]
        report('g5[bf16x3-fwd, auto ce].loss_fused_vs_unfused', b.reshape(1), a.reshape(1), 2e-6)
    finally:
        A.set_precision('bf16')


def test_missing_library_fails_loudly(A, monkeypatch):
    from nuwa_pytorch_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libamdnuwa.so')
    m = A.FeedForward(dim=32).to(DEV)
    with pytest.raises(RuntimeError, match='libamdnuwa'):
        m(torch.randn(1, 4, 32, device=DEV))


def test_training_step_is_bitwise_reproducible(A):
    """no atomics anywhere on the libamdnuwa path: two runs of the same step give a bit-identical loss and bit-identical
    gradients for every decoder-side parameter (the text encoder still runs on torch ops, whose backward is not order-fixed)"""
    torch.manual_seed(3)
    nuwa = _tiny_nuwa(A, False).to(DEV).train()
    g = torch.Generator().manual_seed(9)
    text = torch.randint(1, 50, (3, 8), generator=g).to(DEV)
    vid = torch.randint(0, 64, (3, 3, 4, 4), generator=g).to(DEV)
    vid[:, :, 0, 0] = 7                      # repeated token ids: many rows land on the same embedding row
    runs = []
    for _ in range(2):
        nuwa.zero_grad(set_to_none=True)
        loss = nuwa(text=text, video=vid, return_loss=True, cond_dropout_prob=0.)
        loss.backward()
        runs.append((loss.detach().clone(), {n: p.grad.clone() for n, p in nuwa.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0])
    assert len(runs[0][1]) > 40
    for n, gr in runs[0][1].items():
        if not n.startswith('text_'):
            assert torch.equal(gr, runs[1][1][n]), n


def test_block_chaining_is_bitwise_neutral(A):
    """Transformer.forward_layers hands each block's pre-norm output over from the previous block's post-norm kernel; the loss
    and every gradient must be bit-identical with the hand-over switched off"""
    import nuwa_pytorch_amd.nuwa_pytorch as M
    torch.manual_seed(4)
    nuwa = _tiny_nuwa(A, False).to(DEV).train()
    g = torch.Generator().manual_seed(10)
    text = torch.randint(1, 50, (2, 8), generator=g).to(DEV)
    vid = torch.randint(0, 64, (2, 3, 4, 4), generator=g).to(DEV)
    runs = []
    try:
        for chain in (True, False):
            M.Transformer.chain_blocks = chain
            nuwa.zero_grad(set_to_none=True)
            loss = nuwa(text=text, video=vid, return_loss=True, cond_dropout_prob=0.)
            loss.backward()
            runs.append((loss.detach().clone(), {n: p.grad.clone() for n, p in nuwa.named_parameters() if p.grad is not None}))
    finally:
        M.Transformer.chain_blocks = True
    assert torch.equal(runs[0][0], runs[1][0])
    for n, gr in runs[0][1].items():
        if n.startswith('text_'):
            continue
        if 'norm' in n or n.endswith('to_out.bias'):   # column sums of the LayerNorm backwards: same terms, grouped per workgroup differently by the chained kernel
            report(f'chain_neutral.{n}', gr, runs[1][1][n], 1e-5)
        else:
            assert torch.equal(gr, runs[1][1][n]), n


def test_reversible_stack_memory_is_depth_independent(A):
    """cfg-4 path (dec_reversible=True): with the recomputing backward the peak activation memory of a deep stack stays
    close to that of a shallow one, while the stored-activation mode grows with depth; both give the same gradients"""
    import nuwa_pytorch_amd.nuwa_pytorch as M

    def run(depth, efficient):
        torch.manual_seed(0)
        tr = M.ReversibleTransformer(dim=64, depth=depth, causal=True, heads=2, dim_head=32, cross_attend=True,
                                     sparse_3dna_attn=True, sparse_3dna_kernel_size=3, sparse_3dna_video_shape=(4, 8, 8),
                                     sparse_3dna_dilations=(1, 2), shift_video_tokens=True).to(DEV).train()
        tr.net.memory_efficient = efficient
        g = torch.Generator().manual_seed(1)
        x = torch.randn(16, 256, 64, generator=g).to(DEV).requires_grad_(True)
        ctx = torch.randn(16, 16, 64, generator=g).to(DEV).requires_grad_(True)
        mask = torch.ones(16, 16, dtype=torch.bool, device=DEV)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        y = tr(x, context=ctx, context_mask=mask)
        y.square().mean().backward()
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
        grads = {n: p.grad.clone() for n, p in tr.named_parameters() if p.grad is not None and n.startswith('layers.0.')}
        return peak, x.grad.clone(), ctx.grad.clone(), grads

    A.set_precision('bf16x3')
    try:
        p2, *_ = run(2, True)
        p8, dx_e, dc_e, g_e = run(8, True)
        q2, *_ = run(2, False)
        q8, dx_s, dc_s, g_s = run(8, False)
    finally:
        A.set_precision('bf16')
    print('peak bytes: recompute depth 2 / 8 =', p2, p8, ' stored depth 2 / 8 =', q2, q8)
    # going from depth 2 to depth 8 the stored-activation mode grows by 6 layers of activations; the recomputing mode only by
    # the parameter gradients and cached bf16 weights of those layers
    assert (p8 - p2) < 0.35 * (q8 - q2), (p2, p8, q2, q8)
    assert q8 > 1.5 * p8, (p8, q8)
    report('rev.dx', dx_e, dx_s, 2e-3)
    report('rev.dctx', dc_e, dc_s, 2e-3)
    for n in g_s:
        report(f'rev.grad.{n}', g_e[n], g_s[n], 2e-3)


VA_KW = dict(dim=32, image_size=16, num_audio_tokens=40, num_audio_tokens_per_video_frame=4, max_video_frames=3, text_num_tokens=50,
             text_max_seq_len=8, text_enc_depth=2, text_enc_dim_head=16, text_enc_heads=2, enc_reversible=True, dec_reversible=False,
             dec_depth=3, dec_dim_head=32, dec_heads=2, sparse_3dna_kernel_size=3, sparse_3dna_dilation=2, sparse_2dna_kernel_size=7,
             sparse_2dna_dilation=2, cross_modality_attn_every=3, audio_loss_weight=0.7)


@pytest.mark.parametrize('mode,tol,gtol', MODES)
@pytest.mark.parametrize('name', ['g9a_video_audio', 'g9b_video_audio_relpos', 'g9c_video_audio_reversible'])
def test_g9_video_audio_loss_logits_grads(A, name, mode, tol, gtol):
    """BASELINE cfg 5 (row a15): NUWAVideoAudio with the non-reversible dual decoder, loaded with the reference's state dict,
    against the reference's loss, video / audio logits and every gradient"""
    Ar, P, G = load(name)
    vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
    rev = bool(Ar['reversible']) if 'reversible' in Ar else False
    m = A.NUWAVideoAudio(vae=vae, sparse_3dna_rel_pos_bias=bool(Ar['rel_pos_bias']), **{**VA_KW, 'dec_reversible': rev})
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith('vae.') or '.net.blocks.' in k for k in missing), missing
    m = m.to(DEV).train()
    run_mode(A, mode)
    try:
        text, vid, aud = Ar['text'].to(DEV), Ar['video_ids'].to(DEV), Ar['audio_ids'].to(DEV)
        b = text.shape[0]
        vl, al = m(text=text, video=vid.reshape(b, -1)[:, :-1], audio=aud[:, :-1], return_loss=False, cond_dropout_prob=0.)
        report(f'{name}[{mode}].video_logits', vl, Ar['video_logits'], tol)
        report(f'{name}[{mode}].audio_logits', al, Ar['audio_logits'], tol)
        loss = m(text=text, video=vid, audio=aud, return_loss=True, cond_dropout_prob=0.)
        report(f'{name}[{mode}].loss', loss.reshape(1), Ar['loss'].reshape(1), tol)
        loss.backward()
        # (bf16 mode on the batch-1 fixture: single-sample gradients of the text encoder are the noisiest -> x3)
        n = check_grads(m, G, gtol * (2 if mode == 'bf16x3' else 3), f'{name}[{mode}]', skip=('.net.blocks.',))
        assert n > 150
    finally:
        A.set_precision('bf16')


@pytest.mark.parametrize('mode,tol,gtol', MODES)
def test_g10_text_encoder_on_hip_kernels(A, mode, tol, gtol):
    """row f1: the text encoder's self-attention blocks run on the cross-attention kernels (keys / values = the query rows,
    rotary on q, k and v, padded-key mask) inside the reversible stack; against the reference fixture"""
    import nuwa_pytorch_amd.nuwa_pytorch as M
    Ar, P, G = load('g10_text_encoder')
    tr = M.ReversibleTransformer(dim=32, depth=2, heads=2, dim_head=32)
    missing, unexpected = tr.load_state_dict(P, strict=False)
    assert not unexpected and all('net.blocks.' in k or k.endswith('.mask') for k in missing), (missing, unexpected)
    tr = tr.to(DEV).train()
    assert tr.layers[0][0]._inner(seq_len=12) is not None          # the fused libamdnuwa node is taken, not the torch fallback
    run_mode(A, mode)
    try:
        x = Ar['x'].to(DEV).requires_grad_(True)
        y = tr(x, mask=Ar['mask'].to(DEV), rotary_pos_emb=Ar['freqs'].to(DEV))
        report(f'g10[{mode}].y', y, Ar['y'], tol)
        y.backward(Ar['dy'].to(DEV))
        report(f'g10[{mode}].dx', x.grad, Ar['dx'], gtol)
        assert check_grads(tr, G, gtol * 2, f'g10[{mode}]', skip=('net.blocks.',)) > 20
    finally:
        A.set_precision('bf16')


def _grads_of(mod):
    return {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('n,kernel,dil', [(1, 5, 1), (2, 7, 1), (13, 7, 2), (321, 7, 1)])
def test_sparse_causal_2dna_hip_equals_torch_formulation(A, O_mod, n, kernel, dil):
    """cfg-5 audio window attention: the libamdnuwa route (3DNA kernels on a (time, 1, 1) grid) against the oracle"""
    from nuwa_pytorch_amd.video_audio import SparseCausal2DNA
    torch.manual_seed(0)
    m = SparseCausal2DNA(dim=64, heads=2, dim_head=32, kernel_size=kernel, dilation=dil).to(DEV)
    torch.manual_seed(1)
    x = torch.randn(2, n, 64)
    g = torch.randn(2, n, 64)
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    yr = O_mod.sparse_causal_2dna(xr, P, 2, kernel, dil)
    yr.backward(g)
    A.set_precision('bf16x3')
    try:
        xd = x.to(DEV).requires_grad_(True)
        y = m(xd)
        report(f'audio2dna[{n},{kernel},{dil}].y', y, yr.detach(), 1e-3)
        y.backward(g.to(DEV))
        report(f'audio2dna[{n},{kernel},{dil}].dx', xd.grad, xr.grad, 2e-3)
        if n > 1:
            for k, gr in _grads_of(m).items():
                report(f'audio2dna[{n},{kernel},{dil}].grad.{k}', gr, P[k].grad, 2e-3)
    finally:
        A.set_precision('bf16')


@pytest.fixture(scope='module')
def O_mod():
    from oracle import nuwa_oracle
    return nuwa_oracle


@pytest.mark.parametrize('n_seq,n_ctx,chunk,cchunk', [(1 + 32, 1 + 8, 16, 4), (1 + 30, 1 + 8, 16, 4), (1 + 48, 1 + 6, 16, 4),
                                                      (1 + 8, 1 + 48, 4, 16), (1 + 2560, 1 + 320, 256, 32)])
def test_cross_modality_attention_hip_equals_oracle(A, O_mod, n_seq, n_ctx, chunk, cchunk):
    from nuwa_pytorch_amd.video_audio import CrossModalityCrossAttention
    torch.manual_seed(0)
    m = CrossModalityCrossAttention(dim=64, heads=2, dim_head=32, chunk_size=chunk, context_chunk_size=cchunk).to(DEV)
    with torch.no_grad():
        m.talking_heads.bias.normal_(0, 0.3)
    torch.manual_seed(1)
    x, c, g = torch.randn(2, n_seq, 64), torch.randn(2, n_ctx, 64), torch.randn(2, n_seq, 64)
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xr, cr = x.clone().requires_grad_(True), c.clone().requires_grad_(True)
    yr = O_mod.cross_modality_cross_attention(xr, cr, P, 2, chunk, cchunk)
    yr.backward(g)
    A.set_precision('bf16x3')
    try:
        xd, cd = x.to(DEV).requires_grad_(True), c.to(DEV).requires_grad_(True)
        y = m(xd, cd)
        tag = f'xmodal[{n_seq},{n_ctx}]'
        report(tag + '.y', y, yr.detach(), 1e-3)
        y.backward(g.to(DEV))
        report(tag + '.dx', xd.grad, xr.grad, 2e-3)
        report(tag + '.dctx', cd.grad, cr.grad, 2e-3)
        for k, gr in _grads_of(m).items():
            report(tag + f'.grad.{k}', gr, P[k].grad, 2e-3)
    finally:
        A.set_precision('bf16')


SKETCH_KW = dict(dim=32, image_size=16, max_video_frames=3, sketch_max_video_frames=2, sketch_enc_depth=2, sketch_enc_dim_head=16,
                 sketch_enc_heads=2, dec_depth=3, dec_dim_head=32, dec_heads=2, cross_2dna_kernel_size=3, cross_2dna_dilation=2,
                 sparse_3dna_kernel_size=3, sparse_3dna_dilation=(1, 2))


@pytest.mark.parametrize('mode,tol,gtol', MODES)
@pytest.mark.parametrize('name', ['g11a_sketch', 'g11b_sketch_reversible_3dna'])
def test_g11_sketch_loss_logits_grads(A, name, mode, tol, gtol):
    """row f4: NUWASketch (sketch encoder, SparseCross2DNA decoder) loaded with the reference's state dict, against the
    reference's loss, logits and every gradient.  The fixture carries the token ids of the reference's VAEs, so the tokenizer
    (checked on its own by the g7 tests) is taken out of this comparison."""
    Ar, P, G = load(name)
    extra = dict(enc_reversible=True, dec_reversible=True, sketch_enc_use_sparse_3dna=True) if bool(Ar['reversible']) else {}
    vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
    sketch_vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=48, vq_codebook_dim=32, use_vgg_and_gan=False)
    m = A.NUWASketch(vae=vae, sketch_vae=sketch_vae, **{**SKETCH_KW, **extra})
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith(('vae.', 'sketch_vae.')) or '.net.blocks.' in k for k in missing), missing
    m = m.to(DEV).train()
    sketch_ids = Ar['sketch_ids'].to(DEV)
    m.sketch_vae.get_video_indices = lambda frames: sketch_ids
    run_mode(A, mode)
    try:
        sketch = torch.zeros(2, 2, 3, 16, 16, device=DEV)
        smask = Ar['sketch_mask'].to(DEV) if bool(Ar['has_mask']) else None
        vid = Ar['video_ids'].to(DEV)
        logits = m(sketch=sketch, sketch_mask=smask, video=vid.reshape(2, -1)[:, :-1], return_loss=False, cond_dropout_prob=0.)
        report(f'{name}[{mode}].logits', logits, Ar['logits'], tol)
        loss = m(sketch=sketch, sketch_mask=smask, video=vid, return_loss=True, cond_dropout_prob=0.)
        report(f'{name}[{mode}].loss', loss.reshape(1), Ar['loss'].reshape(1), tol)
        loss.backward()
        n = check_grads(m, G, gtol * 2, f'{name}[{mode}]', skip=('.net.blocks.',))
        assert n > 40
    finally:
        A.set_precision('bf16')


def test_sketch_generate_shapes(A):
    """NUWASketch.generate (np.py:2438-2511): recompute loop on the HIP decoder, greedy sampling is deterministic"""
    torch.manual_seed(5)
    vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
    sketch_vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=48, vq_codebook_dim=32, use_vgg_and_gan=False)
    m = A.NUWASketch(vae=vae, sketch_vae=sketch_vae, **SKETCH_KW).to(DEV).eval()
    sketch = torch.rand(1, 2, 3, 16, 16, generator=torch.Generator().manual_seed(1)).to(DEV)
    A.set_precision('bf16x3')
    try:
        a = m.generate(sketch=sketch, filter_thres=0.99, num_frames=1, cond_scale=2.)
        b = m.generate(sketch=sketch, filter_thres=0.99, num_frames=1, cond_scale=2.)
    finally:
        A.set_precision('bf16')
    assert a.shape == (1, 1, 3, 16, 16) and bool(torch.isfinite(a).all()) and torch.equal(a, b)


def test_reversible_dual_decoder_recomputing_backward(A):
    """cfg 5 default (dec_reversible=True): the recomputing backward of the dual decoder gives the gradients of the
    stored-activation mode (same arithmetic, up to the reconstruction error of x = y - f(.)) with a fraction of its peak memory"""
    import nuwa_pytorch_amd.video_audio as VA

    def run(efficient, depth):
        torch.manual_seed(3)
        dec = VA.ReversibleDualModalityDecoder(dim=64, depth=depth, heads=2, dim_head=32, num_audio_tokens_per_video_frame=8,
                                               num_video_tokens_per_frame=64, sparse_3dna_video_shape=(4, 8, 8),
                                               sparse_3dna_kernel_size=3, sparse_3dna_dilations=(1, 2), shift_video_tokens=True,
                                               shift_audio_tokens=True, cross_modality_attn_every=2).to(DEV)
        g = torch.Generator().manual_seed(4)
        video = torch.randn(2, 1 + 4 * 64, 64, generator=g).to(DEV).requires_grad_(True)
        audio = torch.randn(2, 1 + 4 * 8, 64, generator=g).to(DEV).requires_grad_(True)
        ctx = torch.randn(2, 6, 64, generator=g).to(DEV).requires_grad_(True)
        cmask = torch.ones(2, 6, dtype=torch.bool, device=DEV)
        cmask[1, 4:] = False
        VA.DualModalityReversibleSequence.memory_efficient = efficient
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        v, a = dec(video, audio, context=ctx, context_mask=cmask)
        (v.square().mean() + a.square().mean()).backward()
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
        grads = {n: p.grad.clone() for n, p in dec.named_parameters() if p.grad is not None}
        return peak, grads, video.grad.clone(), audio.grad.clone(), ctx.grad.clone()
    A.set_precision('bf16x3')
    try:
        pe, ge, dve, dae, dce = run(True, 4)
        ps, gs, dvs, das, dcs = run(False, 4)
    finally:
        VA.DualModalityReversibleSequence.memory_efficient = True
        A.set_precision('bf16')
    assert set(ge) == set(gs) and len(ge) > 100
    for n in ge:
        report(f'dual_rev.{n}', ge[n], gs[n], 2e-3)
    report('dual_rev.dvideo', dve, dvs, 2e-3)
    report('dual_rev.daudio', dae, das, 2e-3)
    report('dual_rev.dcontext', dce, dcs, 2e-3)
    assert pe < 0.6 * ps, (pe, ps)


NC_CASES = [((3, 4, 4), 3, 1, 49, 2, 32, False), ((3, 4, 4), 3, 2, 30, 2, 32, True), ((2, 8, 8), (3, 5, 3), 1, 129, 4, 32, False),
            ((2, 16, 16), 3, (1, 2, 2), 400, 8, 64, False), ((4, 16, 16), (5, 3, 3), 2, 1025, 8, 64, True), ((2, 4, 4), 3, 1, 2, 2, 32, False)]


@pytest.mark.parametrize('mode,tol,gtol', MODES)
@pytest.mark.parametrize('shape,kernel,dil,n,heads,dh,rel', NC_CASES)
def test_noncausal_sparse3dna_hip_vs_oracle(A, O_mod, shape, kernel, dil, n, heads, dh, rel, mode, tol, gtol):
    """row f4: the symmetric-window Sparse3DNA of NUWASketch's sketch encoder (causal=False, np.py:429) on the libamdnuwa 3DNA
    kernels, forward + backward against the oracle (itself pinned on the reference in test_sketch_vs_reference.py): whole and partial
    sequences (zero-padded rows ARE attended, score 0), dilation, rel-pos bias, mixed kernel shapes, n = 2"""
    torch.manual_seed(0)
    dim = 64
    m = A.Sparse3DNA(dim=dim, video_shape=shape, kernel_size=kernel, dilation=dil, heads=heads, dim_head=dh, causal=False, rel_pos_bias=rel)
    assert m._hip_ok()
    P = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(n)
    x, dy = torch.randn(2, n, dim, generator=g), torch.randn(2, n, dim, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = O_mod.sparse3dna(xr, P, shape, kernel, dil, heads, causal=False)
    yr.backward(dy)
    m = m.to(DEV)
    run_mode(A, mode)
    try:
        xd = x.to(DEV).requires_grad_(True)
        y = m(xd)
        tag = f'nc3dna[{shape},{kernel},{dil},{n},{mode}]'
        report(tag + '.y', y, yr.detach(), tol)
        y.backward(dy.to(DEV))
        report(tag + '.dx', xd.grad, xr.grad, gtol)
        for k, gr in _grads_of(m).items():
            report(tag + f'.grad.{k}', gr, P[k].grad, gtol)
    finally:
        A.set_precision('bf16')


XC2_CASES = [(4, 2, 32, 3, 1, 2, 1 + 3 * 16, 'frame'), (4, 2, 32, 3, 2, 2, 1 + 21, 'rand'), (8, 4, 32, 5, 1, 1, 1 + 2 * 64, None),
             (16, 8, 64, 3, 1, 2, 1 + 2 * 256, 'rand'), (16, 8, 64, 3, 2, 2, 1 + 300, 'frame'), (4, 2, 32, 3, 1, 2, 1, None)]


@pytest.mark.parametrize('mode,tol,gtol', MODES)
@pytest.mark.parametrize('fmap,heads,dh,kernel,dil,frames,n,masking', XC2_CASES)
def test_sparse_cross_2dna_hip_vs_oracle(A, O_mod, fmap, heads, dh, kernel, dil, frames, n, masking, mode, tol, gtol):
    """row f4: SparseCross2DNA (NUWASketch decoder cross-attention, np.py:761-901) on libamdnuwa: windowed queries on the 3DNA kernels
    pointed at the sketch context, <bos> query glue; forward + backward (x, context, every parameter incl. the null key / value and
    the Conv3d talking heads) against the oracle: whole / partial sequences, masked sketch frames, random key masks, dilation, n = 1"""
    from nuwa_pytorch_amd.nuwa_pytorch import SparseCross2DNA
    torch.manual_seed(0)
    dim = 64
    m = SparseCross2DNA(dim=dim, image_size=fmap, heads=heads, dim_head=dh, kernel_size=kernel, dilation=dil)
    T = frames * fmap * fmap
    assert m._hip_ok(T)
    P = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(n + fmap)
    x, ctx, dy = torch.randn(2, n, dim, generator=g), torch.randn(2, T, dim, generator=g), torch.randn(2, n, dim, generator=g)
    mask = None
    if masking == 'frame':                           # sample 1 hides its last sketch frame (the sketch_mask of NUWASketch.forward)
        mask = torch.ones(2, T, dtype=torch.bool)
        mask[1, (frames - 1) * fmap * fmap:] = False
    elif masking == 'rand':
        mask = torch.rand(2, T, generator=g) > 0.3
    xr, cr = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    yr = O_mod.sparse_cross_2dna(xr, cr, P, heads, fmap, kernel, dil, context_mask=mask)
    yr.backward(dy)
    m = m.to(DEV)
    run_mode(A, mode)
    try:
        xd, cd = x.to(DEV).requires_grad_(True), ctx.to(DEV).requires_grad_(True)
        y = m(xd, context=cd, context_mask=mask.to(DEV) if mask is not None else None)
        tag = f'xc2[{fmap},{heads},{kernel},{dil},{n},{masking},{mode}]'
        report(tag + '.y', y, yr.detach(), tol)
        y.backward(dy.to(DEV))
        report(tag + '.dx', xd.grad, xr.grad, gtol)
        report(tag + '.dctx', cd.grad, cr.grad, gtol)
        for k, gr in _grads_of(m).items():
            if n == 1 and k == 'talking_heads.weight':
                continue                             # only the <bos> query exists: no talking heads on the path
            report(tag + f'.grad.{k}', gr, P[k].grad, gtol)
    finally:
        A.set_precision('bf16')


@pytest.mark.parametrize('mode,frames,expect_hip', [('bf16', 5, True), ('bf16', 6, False), ('bf16x3', 4, True), ('bf16x3', 5, False)])
def test_sparse_cross_2dna_window_at_the_lds_limit(A, O_mod, mode, frames, expect_hip):
    """the window kernels' LDS tables grow with J = frames * kernel^2 + 1 key slots: at fmap 16 / 8 heads x 64 / kernel 5 the backward
    fits the CU's 160 KiB up to J = 126 with bf16 operands and J = 101 with hi + lo pairs.  amdnuwa_s3_supported() says so, _hip_ok
    follows it, and the module gives the oracle's result on EITHER side of the limit (kernels below, PyTorch-op formulation above)
    instead of failing in the backward launch (round-2 advisor finding)."""
    from nuwa_pytorch_amd.nuwa_pytorch import SparseCross2DNA
    fmap, heads, dh, kernel, dim, n = 16, 8, 64, 5, 64, 1 + 200
    torch.manual_seed(0)
    m = SparseCross2DNA(dim=dim, image_size=fmap, heads=heads, dim_head=dh, kernel_size=kernel, dilation=1)
    T = frames * fmap * fmap
    P = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(frames)
    x, ctx, dy = torch.randn(1, n, dim, generator=g), torch.randn(1, T, dim, generator=g), torch.randn(1, n, dim, generator=g)
    xr, cr = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    yr = O_mod.sparse_cross_2dna(xr, cr, P, heads, fmap, kernel, 1)
    yr.backward(dy)
    m = m.to(DEV)
    run_mode(A, mode)
    try:
        assert m._hip_ok(T) == expect_hip
        tol, gtol = (1e-3, 2e-3) if (mode == 'bf16x3' or not expect_hip) else (2e-2, 7e-2)
        xd, cd = x.to(DEV).requires_grad_(True), ctx.to(DEV).requires_grad_(True)
        y = m(xd, context=cd)
        tag = f'xc2_lds_limit[{mode},{frames}]'
        report(tag + '.y', y, yr.detach(), tol)
        y.backward(dy.to(DEV))
        report(tag + '.dx', xd.grad, xr.grad, gtol)
        report(tag + '.dctx', cd.grad, cr.grad, gtol)
        for k, gr in _grads_of(m).items():
            report(tag + f'.grad.{k}', gr, P[k].grad, gtol)
    finally:
        A.set_precision('bf16')
