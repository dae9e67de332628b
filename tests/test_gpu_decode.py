"""Key/value-cached decoding (nuwa_pytorch_amd/decode.py, csrc/decode.hip) on the MI355X.

The decoder is causal, so row t of ONE full forward over a sequence is what the cached decoder must produce at step t when it
is fed the same tokens (teacher forcing).  That pins the cached path to the reference's golden logits (fixture g5) and to the
full-sequence kernels; generate() itself is checked against the reference's recompute loop under greedy sampling."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load  # noqa: E402
from gpu_util import bf_value, report, to_bf_pair  # noqa: E402
from test_gpu_modules import _tiny_nuwa  # noqa: E402

DEV = 'cuda'


@pytest.fixture(scope='module')
def A():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import nuwa_pytorch_amd
    return nuwa_pytorch_amd


@pytest.fixture(scope='module')
def K(A):
    from nuwa_pytorch_amd import kernels
    return kernels


@pytest.fixture(scope='module')
def O():
    from oracle import nuwa_oracle
    return nuwa_oracle


S3_CASES = [((3, 4, 4), (3, 3, 3), (1, 1, 1), 2, 32, True), ((3, 4, 4), (3, 3, 3), (2, 2, 2), 8, 64, False),
            ((2, 8, 8), (5, 3, 3), (1, 2, 4), 4, 32, True), ((4, 4, 4), (3, 5, 3), (1, 1, 2), 8, 64, True)]


@pytest.mark.parametrize('case', range(len(S3_CASES)))
@pytest.mark.parametrize('x3', [False, True])
def test_s3_decode_rows_equal_full_attention(K, O, case, x3):
    """the single-query kernel fed row by row reproduces every row of Sparse3DNA's core (oracle restatement of np.py:488-608),
    with and without the relative-position bias"""
    shape, kern, dil, heads, dh, use_rel = S3_CASES[case]
    n = 1 + shape[0] * shape[1] * shape[2] - 1
    B, inner = 2, heads * dh
    J = kern[0] * kern[1] * kern[2] + 1
    torch.manual_seed(40 + case)
    qkv = torch.randn(B, n, 3, heads, dh)
    if not x3:
        qkv = qkv.bfloat16().float()
    wth = torch.randn(heads, heads) * 0.5 + torch.eye(heads)
    rel = torch.cat((torch.zeros(1, heads), torch.randn(J - 1, heads)), 0) if use_rel else None
    idx = O.neighbor_table(shape, kern, dil, causal=True)
    o_ref = O.sparse3dna_core(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], wth, idx, dh ** -0.5,
                              rel_pos_bias=rel[1:].t().contiguous() if use_rel else None)
    g = K.s3_geom(B, n, shape, kern, dil, heads, dh)
    cache = K.zeros_bf((B, n, 2 * inner), DEV, lo=x3)
    pos = torch.zeros(1, dtype=torch.int32, device=DEV)
    rows = []
    flat = qkv.reshape(B, n, 3 * inner).to(DEV)
    for t in range(n):
        pos.fill_(t)
        o = K.s3_decode(g, to_bf_pair(flat[:, t].contiguous(), x3), cache, pos, wth.to(DEV), rel.to(DEV) if use_rel else None)
        rows.append(bf_value(o))
    got = torch.stack(rows, 1).reshape(B, n, heads, dh)
    report(f's3_decode[{case},x3={x3}]', got, o_ref, 3e-5 if x3 else 2 ** -7)
    # and the cache now holds exactly the key / value rows
    assert torch.equal(cache.hi.reshape(B, n, 2 * inner), to_bf_pair(flat[:, :, inner:].contiguous(), x3).hi)


@pytest.mark.parametrize('x3', [False, True])
def test_decode_shift_rows_equal_shift_video_tokens(K, O, x3):
    torch.manual_seed(5)
    B, fmap, frames, D = 3, 4, 2, 64
    n = 1 + frames * fmap * fmap - 3                    # partial last frame
    h = torch.randn(B, n, D)
    if not x3:
        h = h.bfloat16().float()
    ref = O.shift_video_tokens(h, fmap)
    cache = K.zeros_bf((B, n, D), DEV, lo=x3)
    pos = torch.zeros(1, dtype=torch.int32, device=DEV)
    rows = []
    for t in range(n):
        pos.fill_(t)
        rows.append(bf_value(K.decode_shift(to_bf_pair(h[:, t].contiguous().to(DEV), x3), cache, pos, fmap)))
    report(f'decode_shift[x3={x3}]', torch.stack(rows, 1), ref, 1e-6 if not x3 else 2e-5)


X_CASES = [(3, 7, 2, 32), (2, 33, 8, 64), (2, 256, 8, 64), (4, 31, 4, 32)]


@pytest.mark.parametrize('case', range(len(X_CASES)))
@pytest.mark.parametrize('x3', [False, True])
def test_xattn_decode_equals_cross_attention_core(K, O, case, x3):
    """one query per sample against the packed text keys / values: the oracle's cross-attention core (np.py:339-378) with a
    partially and a fully masked sample"""
    B, T, heads, dh = X_CASES[case]
    inner = heads * dh
    torch.manual_seed(60 + case)
    q = torch.randn(B, 1, heads, dh)
    kv = torch.randn(B, T, 2, heads, dh)
    nk, nv = torch.randn(heads, dh), torch.randn(heads, dh)
    if not x3:
        q, kv, nk, nv = (t.bfloat16().float() for t in (q, kv, nk, nv))
    wth = torch.randn(heads, heads) * 0.5 + torch.eye(heads)
    mask = torch.rand(B, T) > 0.3
    mask[0] = False                                     # condition-dropped sample: only the null key
    o_ref = O.attention_core(q, kv[:, :, 0], kv[:, :, 1], nk, nv, wth, mask, dh ** -0.5)
    g = K.x_geom(B, 1, T, heads, dh)
    pk = K.xattn_pack(g, to_bf_pair(kv.reshape(B * T, 2 * inner).to(DEV), x3), nk.to(DEV), nv.to(DEV),
                      mask.to(torch.uint8).to(DEV))
    o = K.xattn_decode(g, to_bf_pair(q.reshape(B, inner).to(DEV), x3), pk, wth.to(DEV))
    report(f'xattn_decode[{case},x3={x3}]', bf_value(o).reshape(B, 1, heads, dh), o_ref, 3e-5 if x3 else 2 ** -7)


@pytest.mark.parametrize('M,N,Kd', [(1, 512, 512), (4, 1536, 512), (8, 2752, 512), (9, 512, 1376), (16, 8192, 512), (32, 100, 64),
                                    (3, 77, 2752)])
@pytest.mark.parametrize('x3', [False, True])
def test_few_row_gemm(K, M, N, Kd, x3):
    """the weight-streaming NT GEMM the decode step uses (M <= 32 rows): fp32 / bf16 outputs, bias, alpha"""
    torch.manual_seed(M + N)
    a, w, bias = torch.randn(M, Kd), torch.randn(N, Kd), torch.randn(N)
    if not x3:
        a, w = a.bfloat16().float(), w.bfloat16().float()
    ref = (a.double() @ w.double().t()).float()
    ap, wp = to_bf_pair(a.to(DEV), x3), to_bf_pair(w.to(DEV), x3)
    tol = 2e-5 if x3 else 1e-6
    report(f'rows_gemm.f32[{M},{N},{Kd},x3={x3}]', K.gemm_nt(ap, wp), ref, tol)
    report(f'rows_gemm.bias[{M},{N},{Kd},x3={x3}]', K.gemm_nt(ap, wp, bias=bias.to(DEV), alpha=0.5), 0.5 * ref + bias, tol)
    report(f'rows_gemm.bf[{M},{N},{Kd},x3={x3}]', bf_value(K.gemm_nt(ap, wp, out_bf16=True)), ref, tol if x3 else 2 ** -8)


@pytest.mark.parametrize('D,ybf', [(64, False), (512, True), (1024, False)])
@pytest.mark.parametrize('x3', [False, True])
def test_decode_ln_fuses_post_norm_pre_norm_and_shift(K, O, D, ybf, x3):
    """one launch = post-norm + residual, next pre-norm, cache write and shift gather; fed row by row it reproduces
    x + LN(y), and shift(LN(x_new)) over the whole sequence (oracle shift_video_tokens)"""
    import torch.nn.functional as F
    torch.manual_seed(9)
    B, fmap, n = 2, 4, 1 + 2 * 16 - 5
    y, resid = torch.randn(B, n, D) * 1.2, torch.randn(B, n, D)
    if ybf:
        y = y.bfloat16().float()
    w, b, w2, b2 = (torch.randn(D) for _ in range(4))
    x_ref = resid + F.layer_norm(y, (D,), w, b)
    h_ref = O.shift_video_tokens(F.layer_norm(x_ref, (D,), w2, b2), fmap)
    K.set_precision('bf16x3' if x3 else 'bf16')
    try:
        cache = K.zeros_bf((B, n, D), DEV, lo=x3)
        pos = torch.zeros(1, dtype=torch.int32, device=DEV)
        xs, hs = [], []
        for t in range(n):
            pos.fill_(t)
            yt = y[:, t].contiguous().to(DEV)
            yin = K.BF(yt.bfloat16(), None) if ybf else yt
            xn, h = K.decode_ln(yin, resid[:, t].contiguous().to(DEV), (w.to(DEV), b.to(DEV)), (w2.to(DEV), b2.to(DEV)),
                                cache=cache, pos_dev=pos, fmap=fmap)
            xs.append(xn)
            hs.append(bf_value(h))
        report(f'decode_ln.x[{D},x3={x3}]', torch.stack(xs, 1), x_ref, 2e-5)
        report(f'decode_ln.h[{D},x3={x3}]', torch.stack(hs, 1), h_ref, 3e-5 if x3 else 2 ** -7)
        # the two degenerate forms: pre-norm only (first block) and post-norm only (last block)
        _, h0 = K.decode_ln(resid[:, 0].contiguous().to(DEV), None, None, (w2.to(DEV), b2.to(DEV)))
        report(f'decode_ln.pre_only[{D}]', bf_value(h0), F.layer_norm(resid[:, 0], (D,), w2, b2), 3e-5 if x3 else 2 ** -7)
        xl, hl = K.decode_ln(y[:, 1].contiguous().to(DEV), resid[:, 1].contiguous().to(DEV), (w.to(DEV), b.to(DEV)), None)
        assert hl is None
        report(f'decode_ln.post_only[{D}]', xl, x_ref[:, 1], 2e-5)
    finally:
        K.set_precision('bf16')


def _load_g5(A, reversible=False):
    Ar, P, _ = load('g6_nuwa_tiny_reversible' if reversible else 'g5_nuwa_tiny')
    nuwa = _tiny_nuwa(A, reversible)
    nuwa.load_state_dict(P, strict=False)
    return Ar, nuwa.to(DEV).eval()


def _rows_in(nuwa, ids):
    """decoder input rows for token ids [B, m]: <bos>, then embedding + position"""
    with torch.no_grad():
        pos = nuwa.video_pos_emb()
        emb = nuwa.image_embedding(ids) + pos[:ids.shape[1]]
        return torch.cat((nuwa.video_bos[None, None].expand(ids.shape[0], 1, -1), emb), 1)


@pytest.mark.parametrize('graph', [False, True])
@pytest.mark.parametrize('mode,tol', [('bf16x3', 1e-3), ('bf16x3-fwd', 1e-3), ('bf16', 1e-2)])
def test_teacher_forced_cached_logits_match_reference_golden(A, mode, tol, graph):
    """fixture g5 holds the REFERENCE's logits for a 48-token sequence: the cached decoder, fed the same tokens one row at a
    time (eagerly and through the captured HIP graph), must reproduce every row"""
    from nuwa_pytorch_amd.decode import GuidedStepper
    Ar, nuwa = _load_g5(A)
    A.set_precision(mode)
    try:
        with torch.no_grad():
            text = Ar['text'].to(DEV)
            ids = Ar['video_ids'].to(DEV).reshape(2, -1)[:, :-1]
            mask = text != 0
            emb = nuwa.embed_text(text, mask=mask)
            rows = _rows_in(nuwa, ids)
            st = GuidedStepper(nuwa, emb, mask, rows.shape[1], 1., graph=graph)
            got = torch.stack([st(rows[:, t]).clone() for t in range(rows.shape[1])], 1)
        report(f'cached_logits[{mode},graph={graph}]', got, Ar['logits'], tol)
    finally:
        A.set_precision('bf16')


@pytest.mark.parametrize('graph', [False, True])
def test_teacher_forced_cached_logits_reversible_decoder_golden(A, graph):
    """fixture g6 = the REFERENCE's logits with dec_reversible=True (ReversibleTransformer: two residual halves, output = their
    sum): the cached row program must reproduce every row"""
    from nuwa_pytorch_amd.decode import GuidedStepper
    Ar, nuwa = _load_g5(A, reversible=True)
    A.set_precision('bf16x3')
    try:
        with torch.no_grad():
            text = Ar['text'].to(DEV)
            ids = Ar['video_ids'].to(DEV).reshape(2, -1)[:, :-1]
            mask = text != 0
            emb = nuwa.embed_text(text, mask=mask)
            rows = _rows_in(nuwa, ids)
            st = GuidedStepper(nuwa, emb, mask, rows.shape[1], 1., graph=graph)
            got = torch.stack([st(rows[:, t]).clone() for t in range(rows.shape[1])], 1)
        report(f'cached_logits_reversible[graph={graph}]', got, Ar['logits'], 1e-3)
    finally:
        A.set_precision('bf16')


@pytest.mark.parametrize('graph', [False, True])
def test_guided_step_matches_recompute_loop(A, graph):
    """classifier-free guidance as the reference does it (np.py:1894-1898: the final-normed conditioned output is the input of
    the text-masked pass): cached rows vs the full recompute on the same kernels"""
    from nuwa_pytorch_amd.decode import GuidedStepper
    Ar, nuwa = _load_g5(A)
    A.set_precision('bf16x3')
    try:
        with torch.no_grad():
            text = Ar['text'].to(DEV)
            ids = Ar['video_ids'].to(DEV).reshape(2, -1)[:, :20]
            mask = text != 0
            emb = nuwa.embed_text(text, mask=mask)
            rows = _rows_in(nuwa, ids)
            hidden = nuwa.decode_hidden(rows, emb, mask)
            logits = nuwa._final(hidden)
            un = nuwa._final(nuwa.decode_hidden(nuwa.video_transformer.norm(hidden), emb, torch.zeros_like(mask)))
            ref = un + (logits - un) * 2.5
            st = GuidedStepper(nuwa, emb, mask, rows.shape[1], 2.5, graph=graph)
            got = torch.stack([st(rows[:, t]).clone() for t in range(rows.shape[1])], 1)
        report(f'guided_cached_logits[graph={graph}]', got, ref, 1e-3)
    finally:
        A.set_precision('bf16')


@pytest.mark.parametrize('reversible', [False, True])
def test_generate_cached_equals_recompute_under_greedy_sampling(A, reversible):
    """NUWA.generate with the key/value cache and with the reference's recompute loop choose the same tokens when the sampler is
    greedy (filter_thres keeps one logit), hence decode to the same video; shapes follow np.py:1912-1915"""
    torch.manual_seed(12)
    nuwa = _tiny_nuwa(A, reversible).to(DEV).eval()
    text = torch.randint(1, 50, (2, 8), generator=torch.Generator().manual_seed(3)).to(DEV)
    A.set_precision('bf16x3')
    outs = []
    try:
        for cached in (True, False):
            type(nuwa).generate_use_cache = cached
            torch.manual_seed(0)
            outs.append(nuwa.generate(text=text, filter_thres=0.99, num_frames=2, cond_scale=2.))
    finally:
        type(nuwa).generate_use_cache = True
        A.set_precision('bf16')
    assert outs[0].shape == (2, 2, 3, 16, 16)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('reversible', [False, True])
def test_video_audio_generate_alternates_modalities(A, reversible):
    """NUWAVideoAudio.generate (np.py:2111-2222): video and audio tokens sampled alternately a frame at a time on the libamdnuwa
    path (plain and reversible dual decoder): shapes, token ranges, finite frames, determinism under greedy sampling, and the
    guided (two-pass) variant"""
    from test_gpu_modules import VA_KW
    torch.manual_seed(21)
    vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
    m = A.NUWAVideoAudio(vae=vae, sparse_3dna_rel_pos_bias=False, **{**VA_KW, 'dec_reversible': reversible}).to(DEV).eval()
    text = torch.randint(1, 40, (1, 6), generator=torch.Generator().manual_seed(2)).to(DEV)
    A.set_precision('bf16x3')
    try:
        video, audio = m.generate(text=text, filter_thres=0.99, cond_scale=1., num_frames=2)
        tpf, apf = m.num_video_tokens_per_frame, m.num_audio_tokens_per_video_frame
        assert video.shape == (1, 2, 3, 16, 16) and audio.shape == (1, 2 * apf) and audio.dtype == torch.long
        assert int(audio.min()) >= 0 and int(audio.max()) < VA_KW['num_audio_tokens']
        assert bool(torch.isfinite(video).all())
        again, audio2 = m.generate(text=text, filter_thres=0.99, cond_scale=1., num_frames=2)
        assert torch.equal(audio, audio2) and torch.equal(video, again)          # greedy: deterministic
        guided, _ = m.generate(text=text, filter_thres=0.99, cond_scale=2., num_frames=1)
        assert guided.shape == (1, 1, 3, 16, 16) and bool(torch.isfinite(guided).all())
    finally:
        A.set_precision('bf16')


def _tiny_va(A, **over):
    from test_gpu_modules import VA_KW
    torch.manual_seed(21)
    vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
    return A.NUWAVideoAudio(vae=vae, sparse_3dna_rel_pos_bias=True, **{**VA_KW, 'dec_reversible': False, **over}).to(DEV).eval()


@pytest.mark.parametrize('reversible', [False, True])
@pytest.mark.parametrize('shift', [True, False])
def test_dual_decoder_cached_rows_match_full_sequences(A, shift, reversible):
    """DualIncrementalDecoder (one new row per call, two linked streams) against the dual decoder run over the complete sequences:
    rows fed in the sampler's order (start tokens, then one video frame / one audio frame alternately) must reproduce the hidden
    rows of DualModalityDecoder.forward_layers -- 3DNA + audio window caches, both token shifts, text cross-attention, and the
    one-frame-lagged video <-> audio attention with its Conv3d bias"""
    from nuwa_pytorch_amd.decode import DualIncrementalDecoder
    m = _tiny_va(A, shift_video_tokens=shift, shift_audio_tokens=shift, dec_depth=6, dec_reversible=reversible)
    with torch.no_grad():                                  # (fresh modules start with a zero Conv3d bias / tap bias: make them count)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv3d) and mod.bias is not None:
                mod.bias.normal_(0, 0.3)
    gen = torch.Generator().manual_seed(4)
    tpf, apf, F_ = m.num_video_tokens_per_frame, m.num_audio_tokens_per_video_frame, 3
    text = torch.randint(1, 50, (2, 8), generator=gen).to(DEV)
    text[1, 5:] = 0
    vids = torch.randint(0, 64, (2, F_ * tpf), generator=gen).to(DEV)
    aids = torch.randint(0, 40, (2, F_ * apf), generator=gen).to(DEV)
    A.set_precision('bf16x3')
    try:
        with torch.no_grad():
            mask = text != 0
            emb = m.embed_text(text, mask=mask)
            v_in, a_in = m.embed_video(vids), m.embed_audio(aids).contiguous()
            dec = m.video_audio_transformer
            v_ref, a_ref = dec.forward_layers(v_in, a_in, context=emb, context_mask=mask)
            d = DualIncrementalDecoder(dec, 2, v_in.shape[1], a_in.shape[1], emb, mask)
            v_got, a_got = [d.step('v', v_in[:, 0])], [d.step('a', a_in[:, 0])]
            for f in range(F_):
                for t in range(f * tpf, (f + 1) * tpf):
                    v_got.append(d.step('v', v_in[:, 1 + t]))
                for t in range(f * apf, (f + 1) * apf):
                    a_got.append(d.step('a', a_in[:, 1 + t]))
        report(f'dual_cached_video_rows[shift={shift},rev={reversible}]', torch.stack(v_got, 1), v_ref, 1e-4)
        report(f'dual_cached_audio_rows[shift={shift},rev={reversible}]', torch.stack(a_got, 1), a_ref, 1e-4)
    finally:
        A.set_precision('bf16')


@pytest.mark.parametrize('reversible', [False, True])
@pytest.mark.parametrize('cond_scale', [1., 2.])
def test_video_audio_generate_cached_equals_recompute_under_greedy_sampling(A, cond_scale, reversible):
    """NUWAVideoAudio.generate with the per-layer caches and with the reference's recompute loop (np.py:2143-2207) choose the same
    video and audio tokens when the sampler is greedy"""
    m = _tiny_va(A, dec_reversible=reversible)
    text = torch.randint(1, 40, (2, 6), generator=torch.Generator().manual_seed(2)).to(DEV)
    A.set_precision('bf16x3')
    outs = []
    try:
        for cached, graph in ((True, True), (True, False), (False, False)):
            type(m).generate_use_cache, type(m).generate_use_graph = cached, graph
            torch.manual_seed(0)
            outs.append(m.generate(text=text, filter_thres=0.99, cond_scale=cond_scale, num_frames=2))
    finally:
        type(m).generate_use_cache = type(m).generate_use_graph = True
        A.set_precision('bf16')
    (vg, ag), (v0, a0), (v1, a1) = outs
    assert v0.shape == (2, 2, 3, 16, 16) and a0.shape == (2, 2 * m.num_audio_tokens_per_video_frame)
    assert torch.equal(a0, a1) and torch.equal(v0, v1)
    assert torch.equal(ag, a1) and torch.equal(vg, v1)          # the ordinary rows replayed from the two captured HIP graphs


@pytest.mark.parametrize('cached', [True, 'eager', False])
@pytest.mark.parametrize('name', ['g13a_generate_nuwa', 'g13b_generate_nuwa_reversible', 'g13c_generate_video_audio',
                                  'g13d_generate_video_audio_reversible'])
def test_generate_reproduces_the_reference_token_ids(A, name, cached):
    """fixtures g13 hold the token ids the REFERENCE's own generate() sampled (np.py:1841-1915, 2111-2222; greedy: filter_thres 0.99
    keeps one logit) for tiny models with recorded parameters: the key/value-cached row program and the recompute loop on the HIP
    kernels must both sample exactly those video (and audio) tokens"""
    from test_gpu_modules import VA_KW
    Ar, P, _ = load(name)
    rev, cs = bool(Ar['reversible']), float(Ar['cond_scale'])
    if 'audio_ids' in Ar:
        vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
        m = A.NUWAVideoAudio(vae=vae, sparse_3dna_rel_pos_bias=False, **{**VA_KW, 'dec_reversible': rev})
    else:
        m = _tiny_nuwa(A, rev)
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected, unexpected
    m = m.to(DEV).eval()
    text = Ar['text'].to(DEV)
    A.set_precision('bf16x3')
    try:
        type(m).generate_use_cache, type(m).generate_use_graph = bool(cached), cached is True       # 'eager': cached rows, no HIP graph
        torch.manual_seed(0)
        out = m.generate(text=text, filter_thres=0.99, cond_scale=cs, num_frames=2)
    finally:
        type(m).generate_use_cache = type(m).generate_use_graph = True
        A.set_precision('bf16')
    assert torch.equal(m.last_generated_ids.cpu(), Ar['video_ids'].long()), (m.last_generated_ids.cpu(), Ar['video_ids'])
    if 'audio_ids' in Ar:
        assert torch.equal(out[1].cpu(), Ar['audio_ids'].long())


@pytest.mark.parametrize('mode', ['recompute', 'cached', 'cached+graph'])
def test_sketch_generate_reproduces_the_reference_token_ids(A, mode):
    """fixture g13e: the token ids the reference's NUWASketch.generate (np.py:2438-2511) samples (greedy, guided) for a tiny model with
    recorded parameters; the sketch token ids of the reference's randomly initialised sketch VAE are part of the fixture.  Reproduced by
    the reference's recompute algorithm and by the row-at-a-time decoder (SparseCross2DNA rows through decode._Cross2DNARows), eager and
    as a captured HIP graph"""
    from test_gpu_modules import SKETCH_KW
    Ar, P, _ = load('g13e_generate_sketch')
    vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
    sketch_vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=48, vq_codebook_dim=32, use_vgg_and_gan=False)
    m = A.NUWASketch(vae=vae, sketch_vae=sketch_vae, **SKETCH_KW)
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected, unexpected
    m = m.to(DEV).eval()
    sketch_ids = Ar['sketch_ids'].to(DEV)
    m.sketch_vae.get_video_indices = lambda frames: sketch_ids
    A.set_precision('bf16x3')
    try:
        type(m).generate_use_cache, type(m).generate_use_graph = mode != 'recompute', mode == 'cached+graph'
        torch.manual_seed(0)
        frames = m.generate(sketch=Ar['sketch'].to(DEV), filter_thres=0.99, cond_scale=float(Ar['cond_scale']), num_frames=2)
    finally:
        type(m).generate_use_cache = type(m).generate_use_graph = True
        A.set_precision('bf16')
    assert frames.shape == (2, 2, 3, 16, 16)
    assert torch.equal(m.last_generated_ids.cpu(), Ar['video_ids'].long()), (m.last_generated_ids.cpu(), Ar['video_ids'])


@pytest.mark.parametrize('cond_scale', [1., 2.5])
def test_sketch_cached_rows_match_the_recomputed_prefix(A, cond_scale):
    """NUWASketch decoding row by row (decode.GuidedStepper over a decoder with SparseCross2DNA blocks; a sketch mask that hides the
    last sketch frame of one sample) against the logits of the whole recomputed prefix (`_guided_last_logits`, the reference's
    algorithm on the training kernels) at every position of 1.5 frames -- <bos> row, frame border and padding windows included"""
    from test_gpu_modules import SKETCH_KW
    from nuwa_pytorch_amd.decode import GuidedStepper
    torch.manual_seed(21)
    vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=64, vq_codebook_dim=32, use_vgg_and_gan=False)
    sketch_vae = A.VQGanVAE(dim=32, image_size=16, num_layers=2, vq_codebook_size=48, vq_codebook_dim=32, use_vgg_and_gan=False)
    m = A.NUWASketch(vae=vae, sketch_vae=sketch_vae, **SKETCH_KW).to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    sketch = torch.rand(2, 2, 3, 16, 16, generator=g).to(DEV)
    smask = torch.tensor([[True, True], [True, False]], device=DEV)
    tpf, total = 16, 24
    ids = torch.randint(0, 64, (2, total), generator=g).to(DEV)
    A.set_precision('bf16x3')
    try:
        with torch.no_grad():
            ctx, cmask = m.embed_sketch(sketch, mask=smask)
            st = GuidedStepper(m, ctx, cmask, total, cond_scale, graph=False)
            assert sum(b.kind == 'xc2' for b in st.cond.blocks) == SKETCH_KW['dec_depth'] and st.cond.bos_row_differs
            pos_table = m.video_pos_emb()
            row = m.video_bos[None].expand(2, -1)
            worst = 0.
            for t in range(total):
                got = st(row)
                ref = m._guided_last_logits(ids[:, :t], ctx, cmask, cond_scale)
                worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
                row = m.image_embedding(ids[:, t]) + pos_table[t]
    finally:
        A.set_precision('bf16')
    from gpu_util import record
    record(f'sketch_cached_rows[cond_scale={cond_scale}].logits', worst, worst, 1e-3)
    assert worst < 1e-3, worst
