"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU fp32 restatement (plain PyTorch, own formulation) of the NUWA video-decoder
training hot path of lucidrains/nuwa-pytorch.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this file; `nuwa_pytorch_amd/` never does.

Every function cites the reference file:line it restates (paths relative to the
reference checkout; np.py = nuwa_pytorch/nuwa_pytorch.py, vq.py = nuwa_pytorch/vqgan_vae.py,
rev.py = nuwa_pytorch/reversible.py).

Pinning: the reference ships no tests / golden vectors.  The
restatement is pinned against the reference ITSELF, imported read-only in the build
container (tests/test_oracle_vs_reference.py; only runs where /root/reference exists) and
through fixtures generated from that import (tests/golden/*.npz, made by
tests/golden/make_golden.py).  The one boundary that cannot be pinned is the third-party
`vector_quantize_pytorch.VectorQuantize` (not vendored, not installed): `vq_eval_lookup`
below restates its documented eval-path algorithm -- "parity unpinned" at that boundary.

Parameters are passed as a flat dict using the reference's state_dict key names, so a
reference module's `state_dict()` feeds these functions directly.
"""
import math
import torch
import torch.nn.functional as F

FP32_NEG_MAX = -torch.finfo(torch.float32).max


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------

def _tup3(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


def sub(P, prefix):
    """select the sub-dict of P under `prefix.` (prefix stripped)"""
    pl = prefix + '.'
    return {k[len(pl):]: v for k, v in P.items() if k.startswith(pl)}


def layer_norm(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last dim (np.py:120-121 prenorm / postnorm)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def stable_layer_norm(x, w, b):
    """StableLayerNorm np.py:88-95: divide by (detached) row amax, no abs, then LN."""
    x = x / x.amax(dim=-1, keepdim=True).detach()
    return layer_norm(x, w, b)


# --------------------------------------------------------------------------------------
# a5  ShiftVideoTokens (np.py:185-253), shift_space=True / shift_time=False only
# --------------------------------------------------------------------------------------

def shift_video_tokens(x, fmap):
    """x (b, n, D): row 0 = bos (untouched); row 1+p = token at raster position p of a
    (f, fmap, fmap) grid.  Channel chunks follow torch.chunk(4) (ceil-sized chunks):
    chunk0 takes its value from (f, y-1, w) (0 at y==0), chunk1 from (f, y, w-1) (0 at
    w==0), the rest unchanged.  np.py:227, 234-235."""
    b, n, D = x.shape
    if n == 1:
        return x
    c = -(-D // 4)  # chunk(4) -> ceil-sized chunks
    p = torch.arange(n - 1)
    y = (p // fmap) % fmap
    w = p % fmap
    xv = x[:, 1:]
    out = xv.clone()
    # chunk 0: from row p - fmap if y > 0 else 0
    src_h = (p - fmap).clamp(min=0)
    val_h = xv[:, src_h, 0:c] * (y > 0).to(x.dtype)[None, :, None]
    out[:, :, 0:c] = val_h
    # chunk 1: from row p - 1 if w > 0 else 0
    c1 = min(2 * c, D)
    if c1 > c:
        src_w = (p - 1).clamp(min=0)
        val_w = xv[:, src_w, c:c1] * (w > 0).to(x.dtype)[None, :, None]
        out[:, :, c:c1] = val_w
    return torch.cat((x[:, :1], out), dim=1)


# --------------------------------------------------------------------------------------
# a7  neighbour table == what unfoldNd + causal padding computes (np.py:420-457, 507-528)
# --------------------------------------------------------------------------------------

def neighbor_table(video_shape, kernel_size, dilation, causal=True):
    """idx (N, K) int64: for query raster position p, tap slot t=(a*kh+b)*kw+c ->
    key raster position, or -1 when the tap falls in the zero padding (masked).
    causal: taps reach backwards only (padding all on the low side, np.py:427)
    non-causal: symmetric 'same' padding (np.py:429)."""
    Fr, H, W = video_shape
    kf, kh, kw = _tup3(kernel_size)
    df, dh, dw = _tup3(dilation)
    f = torch.arange(Fr)[:, None, None]
    y = torch.arange(H)[None, :, None]
    w = torch.arange(W)[None, None, :]
    cols = []
    for a in range(kf):
        for b_ in range(kh):
            for c in range(kw):
                if causal:
                    ff = f - (kf - 1 - a) * df
                    yy = y - (kh - 1 - b_) * dh
                    ww = w - (kw - 1 - c) * dw
                else:
                    ff = f + (a - (kf - 1) // 2) * df
                    yy = y + (b_ - (kh - 1) // 2) * dh
                    ww = w + (c - (kw - 1) // 2) * dw
                ok = (ff >= 0) & (ff < Fr) & (yy >= 0) & (yy < H) & (ww >= 0) & (ww < W)
                pos = (ff * H + yy) * W + ww
                pos = torch.where(ok, pos, torch.full_like(pos, -1))
                cols.append(pos.reshape(-1))
    return torch.stack(cols, dim=1)


# --------------------------------------------------------------------------------------
# a6  Sparse3DNA (np.py:381-613), causal
# --------------------------------------------------------------------------------------

def sparse3dna_core(q, k, v, w_th, idx, scale, rel_pos_bias=None):
    """The attention core between the projections (np.py:488-608).
    q, k, v: (b, n, h, d) fp32, unscaled q.  w_th: (h, h) talking-heads weight.
    idx: (N, K) neighbour table.  Returns o (b, n, h, d).
    Row 0 is <bos>: its output is v[:,0] (np.py:499, 608)."""
    b, n, h, d = q.shape
    if n == 1:
        return v.clone()
    nq = n - 1
    K = idx.shape[1]
    tab = idx[:nq]                               # (nq, K)
    valid = tab >= 0
    gidx = tab.clamp(min=0) + 1                  # row index into k / v (row 0 is bos)
    need = int(gidx.max()) + 1                   # non-causal windows reach positions the sequence has not filled: the reference
    if k.shape[1] < need:                        # pads them with zero rows, which ARE attended (score 0) -- np.py:472-477, 508
        k = F.pad(k, (0, 0, 0, 0, 0, need - k.shape[1]))
        v = F.pad(v, (0, 0, 0, 0, 0, need - v.shape[1]))
    qs = q[:, 1:] * scale                        # np.py:494-498
    kg = k[:, gidx.reshape(-1)].reshape(b, nq, K, h, d)
    vg = v[:, gidx.reshape(-1)].reshape(b, nq, K, h, d)
    vmask = valid[None, :, :, None, None].to(q.dtype)
    kg = kg * vmask                              # padded taps hold zero k / v (np.py:508)
    vg = vg * vmask
    kb = k[:, :1, None].expand(b, nq, 1, h, d)   # bos key/value first (np.py:532-534)
    vb = v[:, :1, None].expand(b, nq, 1, h, d)
    kk = torch.cat((kb, kg), dim=2)              # (b, nq, J, h, d)
    vv = torch.cat((vb, vg), dim=2)
    sim = torch.einsum('bihd,bijhd->bhij', qs, kk)            # np.py:538
    if rel_pos_bias is not None:                 # (h, K) ; intended per-head broadcast (quirk Q4)
        sim = sim + F.pad(rel_pos_bias, (1, 0))[None, :, None, :]
    mask = F.pad(~valid, (1, 0), value=False)    # bos never masked (np.py:456)
    sim = sim.masked_fill(mask[None, None], FP32_NEG_MAX)     # np.py:548-550
    attn = sim.softmax(dim=-1, dtype=torch.float32)           # np.py:554
    attn = torch.einsum('gh,bhij->bgij', w_th, attn)          # talking heads np.py:556-558
    out = torch.einsum('bgij,bijgd->bigd', attn, vv)          # np.py:564
    return torch.cat((v[:, :1], out), dim=1)                  # np.py:608


def sparse3dna(x, P, video_shape, kernel_size, dilation, heads, idx=None, causal=True):
    """Sparse3DNA.forward np.py:459-613.  P keys: to_q.weight, to_kv.weight,
    talking_heads.weight (h,h,1,1), to_out.weight, to_out.bias [, rel_pos_bias.axial{1,2,3}].
    causal=False (NUWASketch's sketch encoder): symmetric window; row 0 is still treated as <bos> (the reference does)."""
    b, n, D = x.shape
    inner = P['to_q.weight'].shape[0]
    d = inner // heads
    if idx is None:
        idx = neighbor_table(video_shape, kernel_size, dilation, causal=causal)
    q = x @ P['to_q.weight'].t()
    kv = x @ P['to_kv.weight'].t()               # zero pad rows give k=v=0 and are never attended
    k, v = kv[..., :inner], kv[..., inner:]
    if n == 1:                                   # np.py:485-486
        return v @ P['to_out.weight'].t() + P['to_out.bias']
    rpb = None
    if 'rel_pos_bias.axial1' in P:
        kf, kh, kw = _tup3(kernel_size)
        pos = None
        for i, klen in enumerate((kf, kh, kw)):
            if klen <= 1:
                continue
        # AxialPositionalEmbedding(heads, shape=kernel_size) np.py:416, 1693-1709
        axes = [P[f'rel_pos_bias.axial{i + 1}'] for i in range(sum(1 for t in (kf, kh, kw) if t > 1))]
        pos = axes[0]
        for ax in axes[1:]:
            pos = pos.unsqueeze(-2) + ax
        rpb = pos.reshape(-1, heads).t()         # (h, K)
    sh = lambda t: t.reshape(b, n, heads, d)
    o = sparse3dna_core(sh(q), sh(k), sh(v), P['talking_heads.weight'].reshape(heads, heads),
                        idx, d ** -0.5, rpb)
    return o.reshape(b, n, inner) @ P['to_out.weight'].t() + P['to_out.bias']


# --------------------------------------------------------------------------------------
# a8  Attention as cross-attention (np.py:290-379)
# --------------------------------------------------------------------------------------

def attention_core(q, k, v, null_k, null_v, w_th, key_mask, scale, causal=False):
    """q (b,n,h,d); k,v (b,m,h,d); null_k/null_v (h,d); key_mask (b,m) bool or None.
    np.py:339-378."""
    b, n, h, d = q.shape
    nk = null_k[None, None].expand(b, 1, h, d)
    nv = null_v[None, None].expand(b, 1, h, d)
    kk = torch.cat((nk, k), dim=1)
    vv = torch.cat((nv, v), dim=1)
    sim = torch.einsum('bihd,bjhd->bhij', q * scale, kk)
    if key_mask is not None:
        km = F.pad(key_mask, (1, 0), value=True)
        sim = sim.masked_fill(~km[:, None, None, :], FP32_NEG_MAX)
    if causal:
        i, j = sim.shape[-2:]
        cm = torch.ones(i, j, dtype=torch.bool).triu_(j - i + 1)
        sim = sim.masked_fill(cm, FP32_NEG_MAX)
    attn = sim.softmax(dim=-1, dtype=torch.float32)
    attn = torch.einsum('gh,bhij->bgij', w_th, attn)
    return torch.einsum('bgij,bjgd->bigd', attn, vv)


def rotate_half(x):                               # np.py:144-147
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary(freqs, t):                       # np.py:149-153
    rot = freqs.shape[-1]
    t, tp = t[..., :rot], t[..., rot:]
    t = t * freqs.cos() + rotate_half(t) * freqs.sin()
    return torch.cat((t, tp), dim=-1)


def attention(x, P, heads, context=None, context_mask=None, mask=None, rotary=None, causal=False):
    """Attention.forward np.py:315-379 (to_out has no bias)."""
    b, n, D = x.shape
    inner = P['to_q.weight'].shape[0]
    d = inner // heads
    src = context if context is not None else x
    q = (x @ P['to_q.weight'].t()).reshape(b, n, heads, d)
    kv = src @ P['to_kv.weight'].t()
    m = src.shape[1]
    k = kv[..., :inner].reshape(b, m, heads, d)
    v = kv[..., inner:].reshape(b, m, heads, d)
    if context is None and rotary is not None:   # rotary on q, k AND v (quirk Q11) np.py:333-335
        fr = rotary[None, :, None, :]
        q, k, v = (apply_rotary(fr, t) for t in (q, k, v))
    km = context_mask if context is not None else mask
    o = attention_core(q, k, v, P['null_k'].reshape(heads, d), P['null_v'].reshape(heads, d),
                       P['talking_heads.weight'].reshape(heads, heads), km, d ** -0.5, causal)
    return o.reshape(b, n, inner) @ P['to_out.weight'].t()


# --------------------------------------------------------------------------------------
# a9  FeedForward + GEGLU (np.py:255-286)
# --------------------------------------------------------------------------------------

def feedforward(x, P):
    u = x @ P['net.0.weight'].t()
    a, g = u.chunk(2, dim=-1)
    return (a * F.gelu(g)) @ P['net.3.weight'].t()


# --------------------------------------------------------------------------------------
# a3/a4  decoder layer and stack (Transformer np.py:1071-1182)
# --------------------------------------------------------------------------------------

def sandwich(x, P, fn):
    """SandwichNorm np.py:112-128."""
    h = layer_norm(x, P['prenorm.weight'], P['prenorm.bias'])
    h = fn(h)
    return layer_norm(h, P['postnorm.weight'], P['postnorm.bias'])


def decoder_layer(x, P, cfg, layer_idx, context, context_mask):
    """one iteration of Transformer.forward np.py:1174-1180.
    cfg: dict(video_shape, kernel_size, dilations, heads, shift)"""
    fmap = cfg['video_shape'][1]
    dil = cfg['dilations'][layer_idx % len(cfg['dilations'])]
    shift = cfg.get('shift', True)
    key3 = '0.fn.fn' if shift else '0.fn'
    keyf = '2.fn.fn' if shift else '2.fn'
    sh = (lambda t: shift_video_tokens(t, fmap)) if shift else (lambda t: t)
    x = sandwich(x, sub(P, '0'), lambda h: sparse3dna(sh(h), sub(P, key3), cfg['video_shape'],
                                                      cfg['kernel_size'], dil, cfg['heads'])) + x
    x = sandwich(x, sub(P, '1'), lambda h: attention(h, sub(P, '1.fn'), cfg['heads'],
                                                     context=context, context_mask=context_mask)) + x
    x = sandwich(x, sub(P, '2'), lambda h: feedforward(sh(h), sub(P, keyf))) + x
    return x


def decoder_stack(x, P, cfg, context, context_mask):
    """Transformer.forward np.py:1167-1182; P = state dict of `video_transformer`."""
    depth = cfg['depth']
    for l in range(depth):
        x = decoder_layer(x, sub(P, f'layers.{l}'), cfg, l, context, context_mask)
    return stable_layer_norm(x, P['norm.norm.weight'], P['norm.norm.bias'])


def reversible_decoder_stack(x, P, cfg, context, context_mask):
    """ReversibleTransformer np.py:1184-1295 + rev.py:54-142, evaluated in plain (non-memory-
    saving) form: y1 = x1 + f(x2); y2 = x2 + g(y1); out = y1 + y2 halves summed.
    `layers` has 2 entries per depth: [3dna, ff], [cross, ff] (np.py:1246-1277)."""
    fmap = cfg['video_shape'][1]
    shift = cfg.get('shift', True)
    sh = (lambda t: shift_video_tokens(t, fmap)) if shift else (lambda t: t)
    x1, x2 = x, x                                           # rev.py:133
    for l in range(cfg['depth']):
        dil = cfg['dilations'][l % len(cfg['dilations'])]
        A = sub(P, f'layers.{2 * l}')
        # ShiftVideoTokens wrapper is always present in the reversible stack (np.py:1244);
        # with shift_space False it is an identity but still adds a `.fn` level.
        f = lambda h, A=A, dil=dil: sandwich(h, sub(A, '0'), lambda t: sparse3dna(
            sh(t), sub(A, '0.fn.fn'), cfg['video_shape'], cfg['kernel_size'], dil, cfg['heads']))
        g = lambda h, A=A: sandwich(h, sub(A, '1'), lambda t: feedforward(sh(t), sub(A, '1.fn.fn')))
        y1 = x1 + f(x2)
        y2 = x2 + g(y1)
        x1, x2 = y1, y2
        B = sub(P, f'layers.{2 * l + 1}')
        f = lambda h, B=B: sandwich(h, sub(B, '0'), lambda t: attention(
            t, sub(B, '0.fn'), cfg['heads'], context=context, context_mask=context_mask))
        g = lambda h, B=B: sandwich(h, sub(B, '1'), lambda t: feedforward(sh(t), sub(B, '1.fn.fn')))
        y1 = x1 + f(x2)
        y2 = x2 + g(y1)
        x1, x2 = y1, y2
    return stable_layer_norm(x1 + x2, P['norm.norm.weight'], P['norm.norm.bias'])   # rev.py:142


# --------------------------------------------------------------------------------------
# a2  embedding assemble ; a11 logits + CE ; a1 NUWA.forward glue (decoder side)
# --------------------------------------------------------------------------------------

def axial_pos(P, prefix='video_pos_emb'):
    """AxialPositionalEmbedding.forward np.py:1693-1709 (flattened)."""
    pos = None
    i = 1
    while f'{prefix}.axial{i}' in P:
        ax = P[f'{prefix}.axial{i}']
        pos = ax if pos is None else pos.unsqueeze(-2) + ax
        i += 1
    return pos.reshape(-1, pos.shape[-1])


def embed_assemble(ids_in, P, training=True, frac=0.2):
    """np.py:1940-1944 + Embedding np.py:1665-1669 + frac_gradient np.py:83-84.
    ids_in (b, n-1) -> x (b, n, D) with the <bos> row first."""
    emb = P['image_embedding.embed.weight'][ids_in]
    if training and frac < 1:
        emb = emb * frac + emb.detach() * (1 - frac)
    pos = axial_pos(P)
    n1 = ids_in.shape[1]
    x = pos[:n1] + emb
    bos = P['video_bos'][None, None].expand(x.shape[0], 1, -1)
    return torch.cat((bos, x), dim=1)


def decoder_loss(P, cfg, ids, context, context_mask, training=True, return_logits=False):
    """The metric path of NUWA.forward(return_loss=True), decoder side (np.py:1937-1963):
    embed -> video_transformer -> to_logits -> cross entropy.  ids (b, N) int64."""
    x = embed_assemble(ids[:, :-1], P, training=training, frac=cfg.get('embed_frac', 0.2))
    vt = sub(P, 'video_transformer')
    stack = reversible_decoder_stack if cfg.get('reversible', False) else decoder_stack
    h = stack(x, vt, cfg, context, context_mask)
    logits = h @ P['to_logits.weight'].t()
    loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), ids.reshape(-1))
    return (loss, logits) if return_logits else loss


# --------------------------------------------------------------------------------------
# a15  cfg 5: video + audio dual decoder (np.py:157-183, 615-759, 908-1067, 1299-1487, 2224-2293)
# --------------------------------------------------------------------------------------

def shift_audio_tokens(x):
    """ShiftAudioTokens np.py:157-183: first chunk(2) half of the channels comes from the previous token (0 for token 0)."""
    a, b = x.chunk(2, dim=-1)
    return torch.cat((F.pad(a, (0, 0, 1, -1)), b), dim=-1)


def sparse_causal_2dna(x, P, heads, kernel_size, dilation):
    """SparseCausal2DNA np.py:615-759 at height 1.  P: to_qkv.weight, to_out.weight, talking_heads.weight (h,h,1,1,1),
    rel_pos_bias.axial1 (kernel_size, h).  Written with a left-padded key/value tensor and one slice per tap."""
    b, n, _ = x.shape
    q, k, v = (x @ P['to_qkv.weight'].t()).chunk(3, dim=-1)
    if n == 1:
        return v @ P['to_out.weight'].t()
    hd = lambda t: t.reshape(b, n, heads, -1).permute(0, 2, 1, 3)
    q, k, v = hd(q), hd(k), hd(v)
    q = q * q.shape[-1] ** -0.5
    m, pad = n - 1, (kernel_size - 1) * dilation
    kp, vp = F.pad(k[:, :, 1:], (0, 0, pad, 0)), F.pad(v[:, :, 1:], (0, 0, pad, 0))
    t = torch.arange(m)
    sims, vals = [(q[:, :, 1:] * k[:, :, :1]).sum(-1)], [v[:, :, :1].expand(-1, -1, m, -1)]
    bias = P['rel_pos_bias.axial1']                                       # (k, h)
    for a in range(kernel_size):
        ka, va = kp[:, :, a * dilation:a * dilation + m], vp[:, :, a * dilation:a * dilation + m]
        s_a = (q[:, :, 1:] * ka).sum(-1) + bias[a][None, :, None]
        valid = (t - (kernel_size - 1 - a) * dilation) >= 0
        sims.append(s_a.masked_fill(~valid[None, None], -torch.finfo(x.dtype).max))
        vals.append(va)
    attn = torch.stack(sims, dim=-1).softmax(dim=-1, dtype=torch.float32)
    attn = torch.einsum('gh,bhij->bgij', P['talking_heads.weight'].reshape(heads, heads), attn)
    out = (attn[..., None] * torch.stack(vals, dim=-2)).sum(-2)           # b h m d
    out = torch.cat((v[:, :, :1], out), dim=2)
    return out.permute(0, 2, 1, 3).reshape(b, n, -1) @ P['to_out.weight'].t()


def cross_modality_cross_attention(seq, context, P, heads, chunk, ctx_chunk):
    """CrossModalityCrossAttention np.py:908-1067 with has_start_token = context_has_start_token = True, no masks, no inner
    norms: frame f of `seq` (after its start token) attends [null] + frame f of ([0]*(cc-1) + context).  Frame by frame."""
    b, n, d = seq.shape
    body = seq[:, 1:]
    ctx = F.pad(context, (0, 0, ctx_chunk - 1, 0))
    ctx = F.pad(ctx, (0, 0, 0, (-ctx.shape[1]) % ctx_chunk))
    nf = min(-(-body.shape[1] // chunk), ctx.shape[1] // ctx_chunk)
    w_th, b_th = P['talking_heads.weight'].reshape(heads, heads), P['talking_heads.bias']
    outs = []
    for f in range(nf):
        qf = body[:, f * chunk:(f + 1) * chunk]
        qf = F.pad(qf, (0, 0, 0, chunk - qf.shape[1]))
        cf = ctx[:, f * ctx_chunk:(f + 1) * ctx_chunk]
        hd = lambda t: t.reshape(b, t.shape[1], heads, -1).permute(0, 2, 1, 3)
        q = hd(qf @ P['to_q.weight'].t())
        k, v = (hd(t) for t in (cf @ P['to_kv.weight'].t()).chunk(2, dim=-1))
        q = q * q.shape[-1] ** -0.5
        k = torch.cat((P['null_k'][None, :, None].expand(b, -1, -1, -1), k), dim=2)
        v = torch.cat((P['null_v'][None, :, None].expand(b, -1, -1, -1), v), dim=2)
        attn = (q @ k.transpose(-1, -2)).softmax(dim=-1, dtype=torch.float32)
        attn = torch.einsum('gh,bhij->bgij', w_th, attn) + b_th[None, :, None, None]
        outs.append((attn @ v).permute(0, 2, 1, 3).reshape(b, chunk, -1) @ P['to_out.weight'].t())
    out = torch.cat(outs, dim=1) if outs else seq.new_zeros(b, 0, d)
    out = F.pad(out, (0, 0, 0, max(0, body.shape[1] - out.shape[1])))[:, :body.shape[1]]
    return F.pad(out, (0, 0, 1, 0))


def dual_decoder(video, audio, P, cfg, context, context_mask):
    """DualModalityDecoder.forward np.py:1435-1487.  cfg: depth, heads, video_shape, kernel_size, dilations, rel_pos_bias (video),
    audio_kernel, audio_dilations, every, v_per_frame, a_per_frame, shift_video, shift_audio."""
    fmap = cfg['video_shape'][1]
    shv = (lambda t: shift_video_tokens(t, fmap)) if cfg.get('shift_video', True) else (lambda t: t)
    sha = shift_audio_tokens if cfg.get('shift_audio', True) else (lambda t: t)
    kv = '.fn.fn' if cfg.get('shift_video', True) else '.fn'
    ka = '.fn.fn' if cfg.get('shift_audio', True) else '.fn'
    li = 0
    for ind in range(cfg['depth']):
        L = sub(P, f'layers.{li}')
        li += 1
        V, A = sub(L, '0'), sub(L, '1')
        dv = cfg['dilations'][ind % len(cfg['dilations'])]
        da = cfg['audio_dilations'][ind % len(cfg['audio_dilations'])]
        video = sandwich(video, sub(V, '0'), lambda h: sparse3dna(shv(h), sub(V, '0' + kv), cfg['video_shape'], cfg['kernel_size'],
                                                                 dv, cfg['heads'])) + video
        video = sandwich(video, sub(V, '1'), lambda h: attention(h, sub(V, '1.fn'), cfg['heads'], context=context,
                                                                 context_mask=context_mask)) + video
        video = sandwich(video, sub(V, '2'), lambda h: feedforward(shv(h), sub(V, '2' + kv))) + video
        audio = sandwich(audio, sub(A, '0'), lambda h: sparse_causal_2dna(sha(h), sub(A, '0' + ka), cfg['heads'], cfg['audio_kernel'],
                                                                          da)) + audio
        audio = sandwich(audio, sub(A, '1'), lambda h: attention(h, sub(A, '1.fn'), cfg['heads'], context=context,
                                                                 context_mask=context_mask)) + audio
        audio = sandwich(audio, sub(A, '2'), lambda h: feedforward(sha(h), sub(A, '2' + ka))) + audio
        if (ind + 1) % cfg['every'] == 0:
            L = sub(P, f'layers.{li}')
            li += 1
            V, A = sub(L, '0'), sub(L, '1')
            v2 = sandwich(video, sub(V, '0'), lambda h: cross_modality_cross_attention(h, audio, sub(V, '0.fn'), cfg['heads'],
                                                                                        cfg['v_per_frame'], cfg['a_per_frame'])) + video
            a2 = sandwich(audio, sub(A, '0'), lambda h: cross_modality_cross_attention(h, video, sub(A, '0.fn'), cfg['heads'],
                                                                                        cfg['a_per_frame'], cfg['v_per_frame'])) + audio
            video = sandwich(v2, sub(V, '1'), lambda h: feedforward(h, sub(V, '1.fn'))) + v2
            audio = sandwich(a2, sub(A, '1'), lambda h: feedforward(h, sub(A, '1.fn'))) + a2
    return (stable_layer_norm(video, P['video_norm.norm.weight'], P['video_norm.norm.bias']),
            stable_layer_norm(audio, P['audio_norm.norm.weight'], P['audio_norm.norm.bias']))


def reversible_dual_decoder(video, audio, P, cfg, context, context_mask):
    """ReversibleDualModalityDecoder np.py:1489-1655 through DualModalityReversibleSequence (reversible_video_audio.py:27-407),
    evaluated in plain form: streams duplicated into halves, per block y1 = x1 + f(x2), y2 = x2 + g(y1), n1 = m1 + j(m2),
    n2 = m2 + k(n1); the cross-modality block feeds y1 through `k` and n1 through `g` and lets the audio side attend the updated
    video half (reversible_video_audio.py:241-244); halves AVERAGED at the end."""
    fmap = cfg['video_shape'][1]
    shv = (lambda t: shift_video_tokens(t, fmap)) if cfg.get('shift_video', True) else (lambda t: t)
    sha = shift_audio_tokens if cfg.get('shift_audio', True) else (lambda t: t)
    kv = '.fn.fn' if cfg.get('shift_video', True) else '.fn'
    ka = '.fn.fn' if cfg.get('shift_audio', True) else '.fn'
    x1 = x2 = video
    m1 = m2 = audio
    li = 0
    for ind in range(cfg['depth']):
        L = sub(P, f'layers.{li}')
        li += 1
        dv = cfg['dilations'][ind % len(cfg['dilations'])]
        da = cfg['audio_dilations'][ind % len(cfg['audio_dilations'])]
        y1 = x1 + sandwich(x2, sub(L, '0'), lambda h: sparse3dna(shv(h), sub(L, '0' + kv), cfg['video_shape'], cfg['kernel_size'], dv, cfg['heads']))
        y2 = x2 + sandwich(y1, sub(L, '1'), lambda h: feedforward(shv(h), sub(L, '1' + kv)))
        n1 = m1 + sandwich(m2, sub(L, '2'), lambda h: sparse_causal_2dna(sha(h), sub(L, '2' + ka), cfg['heads'], cfg['audio_kernel'], da))
        n2 = m2 + sandwich(n1, sub(L, '3'), lambda h: feedforward(sha(h), sub(L, '3' + ka)))
        x1, x2, m1, m2 = y1, y2, n1, n2
        L = sub(P, f'layers.{li}')
        li += 1
        y1 = x1 + sandwich(x2, sub(L, '0'), lambda h: attention(h, sub(L, '0.fn'), cfg['heads'], context=context, context_mask=context_mask))
        y2 = x2 + sandwich(y1, sub(L, '1'), lambda h: feedforward(h, sub(L, '1.fn')))
        n1 = m1 + sandwich(m2, sub(L, '2'), lambda h: attention(h, sub(L, '2.fn'), cfg['heads'], context=context, context_mask=context_mask))
        n2 = m2 + sandwich(n1, sub(L, '3'), lambda h: feedforward(h, sub(L, '3.fn')))
        x1, x2, m1, m2 = y1, y2, n1, n2
        if (ind + 1) % cfg['every'] == 0:
            L = sub(P, f'layers.{li}')
            li += 1
            y1 = x1 + cross_modality_cross_attention(x2, m2, sub(L, '0'), cfg['heads'], cfg['v_per_frame'], cfg['a_per_frame'])
            y2 = x2 + feedforward(y1, sub(L, '3'))            # the AUDIO FeedForward module (block slot k) on the video stream
            n1 = m1 + cross_modality_cross_attention(m2, y2, sub(L, '2'), cfg['heads'], cfg['a_per_frame'], cfg['v_per_frame'])
            n2 = m2 + feedforward(n1, sub(L, '1'))            # the VIDEO FeedForward module (block slot g) on the audio stream
            x1, x2, m1, m2 = y1, y2, n1, n2
    return (stable_layer_norm((x1 + x2) * 0.5, P['video_norm.norm.weight'], P['video_norm.norm.bias']),
            stable_layer_norm((m1 + m2) * 0.5, P['audio_norm.norm.weight'], P['audio_norm.norm.bias']))


def video_audio_loss(P, cfg, ids, audio_ids, context, context_mask, training=True, return_logits=False):
    """decoder side of NUWAVideoAudio.forward(return_loss=True) np.py:2245-2293; ids (b, N), audio_ids (b, A) int64."""
    fr = cfg.get('embed_frac', 0.2)
    x = embed_assemble(ids[:, :-1], P, training=training, frac=fr)
    ae = P['audio_embedding.embed.weight'][audio_ids[:, :-1]]
    if training and fr < 1:
        ae = ae * fr + ae.detach() * (1 - fr)
    ae = ae + P['audio_pos_emb.axial1'][:ae.shape[1]][None]
    a = torch.cat((P['audio_bos'][None, None].expand(ae.shape[0], 1, -1), ae), dim=1)
    dec = reversible_dual_decoder if cfg.get('reversible', False) else dual_decoder
    v, a = dec(x, a, sub(P, 'video_audio_transformer'), cfg, context, context_mask)
    vl, al = v @ P['to_video_logits.weight'].t(), a @ P['to_audio_logits.weight'].t()
    loss = F.cross_entropy(vl.reshape(-1, vl.shape[-1]), ids.reshape(-1)) + \
        cfg.get('audio_loss_weight', 1.) * F.cross_entropy(al.reshape(-1, al.shape[-1]), audio_ids.reshape(-1))
    return (loss, vl, al) if return_logits else loss


# --------------------------------------------------------------------------------------
# f3  generate(): the reference's decoding algorithm with a greedy sampler (np.py:1841-1915 and 2111-2222)
# --------------------------------------------------------------------------------------

def _guided(cond, uncond, cond_scale):
    return cond if cond_scale == 1 else uncond + (cond - uncond) * cond_scale


def nuwa_generate_greedy(P, cfg, text, num_tokens, cond_scale=1.):
    """NUWA.generate np.py:1870-1908 with arg-max in place of top-k + Gumbel (what filter_thres -> 1 reduces to): every token from a
    pass over the WHOLE prefix; with guidance a second pass is fed the first pass's final-normed OUTPUT rows and sees no text
    (np.py:1894-1898).  Returns ids (b, num_tokens)."""
    ctx, mask = text_encoder(text, P, cfg, training=False)
    vt = sub(P, 'video_transformer')
    stack = reversible_decoder_stack if cfg.get('reversible', False) else decoder_stack
    ids = torch.empty((text.shape[0], 0), dtype=torch.long)
    for _ in range(num_tokens):
        h = stack(embed_assemble(ids, P, training=False), vt, cfg, ctx, mask)
        logits = h[:, -1] @ P['to_logits.weight'].t()
        if cond_scale != 1:
            hu = stack(h, vt, cfg, ctx, torch.zeros_like(mask))
            logits = _guided(logits, hu[:, -1] @ P['to_logits.weight'].t(), cond_scale)
        ids = torch.cat((ids, logits.argmax(-1, keepdim=True)), dim=1)
    return ids


def video_audio_generate_greedy(P, cfg, text, num_frames, cond_scale=1.):
    """NUWAVideoAudio.generate np.py:2143-2207, greedy: video and audio tokens alternately, one video frame's worth at a time, each
    from both decoders run over the whole prefix (the guided second pass takes BOTH normed output streams).  Returns (video ids,
    audio ids)."""
    ctx, mask = text_encoder(text, P, cfg, training=False)
    T = sub(P, 'video_audio_transformer')
    dec = reversible_dual_decoder if cfg.get('reversible', False) else dual_decoder
    b = text.shape[0]
    vids, aids = torch.empty((b, 0), dtype=torch.long), torch.empty((b, 0), dtype=torch.long)
    tv, ta = num_frames * cfg['v_per_frame'], num_frames * cfg['a_per_frame']
    video_turn = True
    while vids.shape[1] < tv or aids.shape[1] < ta:
        x = embed_assemble(vids, P, training=False)
        ae = P['audio_embedding.embed.weight'][aids] + P['audio_pos_emb.axial1'][:aids.shape[1]][None]
        a = torch.cat((P['audio_bos'][None, None].expand(b, 1, -1), ae), dim=1)
        v, au = dec(x, a, T, cfg, ctx, mask)
        W = P['to_video_logits.weight'] if video_turn else P['to_audio_logits.weight']
        logits = (v if video_turn else au)[:, -1] @ W.t()
        if cond_scale != 1:
            v2, a2 = dec(v, au, T, cfg, ctx, torch.zeros_like(mask))
            logits = _guided(logits, (v2 if video_turn else a2)[:, -1] @ W.t(), cond_scale)
        tok = logits.argmax(-1, keepdim=True)
        if video_turn:
            vids = torch.cat((vids, tok), dim=1)
            boundary = vids.shape[1] % cfg['v_per_frame'] == 0
        else:
            aids = torch.cat((aids, tok), dim=1)
            boundary = aids.shape[1] % cfg['a_per_frame'] == 0
        if boundary:
            video_turn = not video_turn
    return vids, aids


# --------------------------------------------------------------------------------------
# f1  text encoder (embed_text np.py:1821-1839; always a ReversibleTransformer in practice, Q1)
# --------------------------------------------------------------------------------------

def rotary_freqs(inv_freq, seq_len):              # np.py:138-142
    t = torch.arange(seq_len).type_as(inv_freq)
    fr = torch.einsum('i,j->ij', t, inv_freq)
    return torch.cat((fr, fr), dim=-1)


def text_encoder(text, P, cfg, training=True):
    """embed_text with enc_reversible=True: blocks (Attention, FeedForward) per depth,
    no ShiftVideoTokens shifting (shift_space False -> identity wrapper level `.fn`)."""
    mask = text != 0
    emb = P['text_embedding.embed.weight'][text]
    if training and cfg.get('embed_frac', 0.2) < 1:
        fr = cfg.get('embed_frac', 0.2)
        emb = emb * fr + emb.detach() * (1 - fr)
    if 'text_rotary_pos_emb.inv_freq' in P:
        rot = rotary_freqs(P['text_rotary_pos_emb.inv_freq'], text.shape[1])
    else:                                      # learned absolute positions (np.py:1827-1829; NUWAVideoAudio's default)
        rot = None
        emb = emb + P['text_abs_pos_emb.embed.weight'][:text.shape[1]][None]
    return text_encoder_stack(emb, sub(P, 'text_transformer'), cfg['text_depth'], cfg['text_heads'], mask, rot), mask


def text_encoder_stack(x, T, depth, heads, mask, rot):
    """ReversibleTransformer of (self-Attention, FeedForward) blocks + final StableLayerNorm (np.py:1184-1295, rev.py:54-142),
    plain (stored-activation) evaluation; T = its state dict (`layers.*` keys)."""
    x1, x2 = x, x
    for l in range(depth):
        A = sub(T, f'layers.{l}')
        f = lambda h, A=A: sandwich(h, sub(A, '0'), lambda t: attention(t, sub(A, '0.fn.fn'), heads, mask=mask, rotary=rot))
        g = lambda h, A=A: sandwich(h, sub(A, '1'), lambda t: feedforward(t, sub(A, '1.fn.fn')))
        y1 = x1 + f(x2)
        y2 = x2 + g(y1)
        x1, x2 = y1, y2
    return stable_layer_norm(x1 + x2, T['norm.norm.weight'], T['norm.norm.bias'])


# --------------------------------------------------------------------------------------
# f4  NUWASketch (np.py:761-901, 2297-2571): SparseCross2DNA, sketch encoder, decoder with 2-D nearby cross-attention
# --------------------------------------------------------------------------------------

def sparse_cross_2dna(x, context, P, heads, image_size, kernel_size, dilation, context_mask=None):
    """SparseCross2DNA.forward np.py:796-901.  Video token at feature-map position i attends a learned null key + the
    kernel_size^2 window around i in EVERY sketch frame; the <bos> row attends null + all sketch tokens, without talking heads."""
    b, n, _ = x.shape
    h = heads
    tpf, kn = image_size ** 2, kernel_size ** 2
    if context_mask is None:
        context_mask = torch.ones(b, context.shape[1], dtype=torch.bool)
    inner = P['to_q.weight'].shape[0]
    q = (x @ P['to_q.weight'].t()).reshape(b, n, h, -1).transpose(1, 2) * (inner // h) ** -0.5
    kv = context @ P['to_kv.weight'].t()
    k, v = (t.reshape(b, -1, h, inner // h).transpose(1, 2) for t in kv.chunk(2, dim=-1))      # b h m d
    nk, nv = P['null_k'][None].expand(b, -1, -1, -1), P['null_v'][None].expand(b, -1, -1, -1)
    sim_bos = torch.einsum('bhd,bhjd->bhj', q[:, :, 0], torch.cat((nk, k), dim=-2))
    sim_bos = sim_bos.masked_fill(~F.pad(context_mask[:, None], (1, 0), value=True), FP32_NEG_MAX)
    out_bos = torch.einsum('bhj,bhjd->bhd', sim_bos.softmax(dim=-1, dtype=torch.float32), torch.cat((nv, v), dim=-2)).reshape(b, 1, -1)
    if n == 1:
        return out_bos @ P['to_out.weight'].t()
    f = context.shape[1] // tpf
    tab = neighbor_table((1, image_size, image_size), (1, kernel_size, kernel_size), (1, dilation, dilation), causal=False)  # (tpf, kn)
    valid = tab >= 0
    idx = tab.clamp(min=0).reshape(-1)
    gather = lambda t: t.reshape(b, h, f, tpf, -1)[:, :, :, idx].reshape(b, h, f, tpf, kn, -1).permute(0, 1, 3, 2, 4, 5).reshape(b, h, tpf, f * kn, -1)
    kf, vf = gather(k), gather(v)                                                         # slot order (frame, tap), np.py:855
    kf = torch.cat((nk[:, :, None].expand(-1, -1, tpf, -1, -1), kf), dim=-2)
    vf = torch.cat((nv[:, :, None].expand(-1, -1, tpf, -1, -1), vf), dim=-2)
    qr = q[:, :, 1:]
    qr = F.pad(qr, (0, 0, 0, (-qr.shape[2]) % tpf)).reshape(b, h, -1, tpf, qr.shape[-1])
    sim = torch.einsum('bhfid,bhijd->bhfij', qr, kf)
    cm = context_mask.reshape(b, f, tpf)[:, :, idx].reshape(b, f, tpf, kn) & valid[None, None]
    cm = F.pad(cm.permute(0, 2, 1, 3).reshape(b, 1, 1, tpf, f * kn), (1, 0), value=True)
    attn = sim.masked_fill(~cm, FP32_NEG_MAX).softmax(dim=-1, dtype=torch.float32)
    attn = torch.einsum('gh,bhfij->bgfij', P['talking_heads.weight'].reshape(h, h), attn)
    out = torch.einsum('bhfij,bhijd->bhfid', attn, vf).permute(0, 2, 3, 1, 4).reshape(b, -1, inner)
    return torch.cat((out_bos, out), dim=1)[:, :n] @ P['to_out.weight'].t()


def sketch_encoder(tokens, T, cfg, mask):
    """`sketch_transformer` (np.py:2346-2360): plain Transformer (self-attention, FF) or, reversible, (attn, FF) blocks; with
    cfg['enc_3dna'] the self-attention is a NON-causal Sparse3DNA under ShiftVideoTokens."""
    shape, heads, fmap = cfg['sketch_shape'], cfg['enc_heads'], cfg['sketch_shape'][1]
    use3, rev = cfg.get('enc_3dna', False), cfg.get('enc_reversible', False)
    shift_on = use3 and cfg.get('shift', True)
    sh = (lambda t: shift_video_tokens(t, fmap)) if shift_on else (lambda t: t)

    def attn_fn(A, key, l):
        if use3:
            dil = cfg['dilations'][l % len(cfg['dilations'])]
            return lambda t: sparse3dna(sh(t), sub(A, key), shape, cfg['kernel_size'], dil, heads, causal=False)
        return lambda t: attention(t, sub(A, key), heads, mask=mask)
    if not rev:
        x = tokens
        wrapped = use3 and cfg.get('shift', True)            # the non-reversible Transformer only wraps when it shifts (np.py:1154-1157)
        for l in range(cfg['enc_depth']):
            A = sub(T, f'layers.{l}')
            x = sandwich(x, sub(A, '0'), attn_fn(A, '0.fn.fn' if wrapped else '0.fn', l)) + x
            x = sandwich(x, sub(A, '2'), lambda t, A=A: feedforward(sh(t), sub(A, '2.fn.fn' if wrapped else '2.fn'))) + x
        return stable_layer_norm(x, T['norm.norm.weight'], T['norm.norm.bias'])
    x1 = x2 = tokens
    for l in range(cfg['enc_depth']):
        A = sub(T, f'layers.{l}')
        y1 = x1 + sandwich(x2, sub(A, '0'), attn_fn(A, '0.fn.fn', l))
        y2 = x2 + sandwich(y1, sub(A, '1'), lambda t, A=A: feedforward(sh(t), sub(A, '1.fn.fn')))
        x1, x2 = y1, y2
    return stable_layer_norm(x1 + x2, T['norm.norm.weight'], T['norm.norm.bias'])


def sketch_decoder(x, V, cfg, context, context_mask):
    """`video_transformer` of NUWASketch: causal 3DNA, SparseCross2DNA over the sketch tokens, FF (np.py:2386-2405)."""
    shape, heads, fmap = cfg['video_shape'], cfg['heads'], cfg['video_shape'][1]
    shift = cfg.get('shift', True)
    sh = (lambda t: shift_video_tokens(t, fmap)) if shift else (lambda t: t)
    cross = lambda A, key, l: (lambda t: sparse_cross_2dna(t, context, sub(A, key), heads, fmap, cfg['cross_kernel'],
                                                           cfg['cross_dilations'][l % len(cfg['cross_dilations'])], context_mask))
    s3 = lambda A, key, l: (lambda t: sparse3dna(sh(t), sub(A, key), shape, cfg['kernel_size'],
                                                 cfg['dilations'][l % len(cfg['dilations'])], heads))
    if not cfg.get('dec_reversible', False):
        for l in range(cfg['depth']):
            A = sub(V, f'layers.{l}')
            x = sandwich(x, sub(A, '0'), s3(A, '0.fn.fn' if shift else '0.fn', l)) + x
            x = sandwich(x, sub(A, '1'), cross(A, '1.fn', l)) + x
            x = sandwich(x, sub(A, '2'), lambda t, A=A: feedforward(sh(t), sub(A, '2.fn.fn' if shift else '2.fn'))) + x
        return stable_layer_norm(x, V['norm.norm.weight'], V['norm.norm.bias'])
    x1 = x2 = x
    for l in range(cfg['depth']):
        A, B = sub(V, f'layers.{2 * l}'), sub(V, f'layers.{2 * l + 1}')
        y1 = x1 + sandwich(x2, sub(A, '0'), s3(A, '0.fn.fn', l))
        y2 = x2 + sandwich(y1, sub(A, '1'), lambda t, A=A: feedforward(sh(t), sub(A, '1.fn.fn')))
        x1, x2 = y1, y2
        y1 = x1 + sandwich(x2, sub(B, '0'), cross(B, '0.fn', l))
        y2 = x2 + sandwich(y1, sub(B, '1'), lambda t, B=B: feedforward(sh(t), sub(B, '1.fn.fn')))
        x1, x2 = y1, y2
    return stable_layer_norm(x1 + x2, V['norm.norm.weight'], V['norm.norm.bias'])


def sketch_loss(P, cfg, sketch_ids, video_ids, sketch_mask=None, training=True):
    """NUWASketch.forward(return_loss=True) from token ids (np.py:2514-2571): returns (loss, logits, sketch_embeds)."""
    b = sketch_ids.shape[0]
    sk = sketch_ids.reshape(b, -1)
    fr = cfg.get('embed_frac', 0.2)
    emb = P['sketch_embedding.embed.weight'][sk]
    if training and fr < 1:
        emb = emb * fr + emb.detach() * (1 - fr)
    tokens = emb + axial_pos(P, 'sketch_pos_emb')[:sk.shape[1]]
    frames = sketch_ids.shape[1]
    mask = torch.ones(b, sk.shape[1], dtype=torch.bool) if sketch_mask is None else \
        sketch_mask[:, :, None].expand(-1, -1, sk.shape[1] // frames).reshape(b, -1)
    ctx = sketch_encoder(tokens, sub(P, 'sketch_transformer'), cfg, mask)
    ids = video_ids.reshape(b, -1)
    x = embed_assemble(ids[:, :-1], P, training=training, frac=fr)
    h = sketch_decoder(x, sub(P, 'video_transformer'), cfg, ctx, mask)
    logits = h @ P['to_logits.weight'].t()
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), ids.reshape(-1)), logits, ctx


def sketch_generate_greedy(P, cfg, sketch_ids, num_tokens, cond_scale=1., sketch_mask=None):
    """NUWASketch.generate np.py:2438-2511 from sketch token ids, greedy: as nuwa_generate_greedy with the sketch encoder's output as
    the context of the SparseCross2DNA decoder; the guided second pass sees a fully masked sketch."""
    b = sketch_ids.shape[0]
    sk = sketch_ids.reshape(b, -1)
    tokens = P['sketch_embedding.embed.weight'][sk] + axial_pos(P, 'sketch_pos_emb')[:sk.shape[1]]
    frames = sketch_ids.shape[1]
    mask = torch.ones(b, sk.shape[1], dtype=torch.bool) if sketch_mask is None else \
        sketch_mask[:, :, None].expand(-1, -1, sk.shape[1] // frames).reshape(b, -1)
    ctx = sketch_encoder(tokens, sub(P, 'sketch_transformer'), cfg, mask)
    V = sub(P, 'video_transformer')
    ids = torch.empty((b, 0), dtype=torch.long)
    for _ in range(num_tokens):
        h = sketch_decoder(embed_assemble(ids, P, training=False), V, cfg, ctx, mask)
        logits = h[:, -1] @ P['to_logits.weight'].t()
        if cond_scale != 1:
            hu = sketch_decoder(h, V, cfg, ctx, torch.zeros_like(mask))
            logits = _guided(logits, hu[:, -1] @ P['to_logits.weight'].t(), cond_scale)
        ids = torch.cat((ids, logits.argmax(-1, keepdim=True)), dim=1)
    return ids


# --------------------------------------------------------------------------------------
# a13  VQGanVAE encode path (vq.py:431-435) -- frozen tokenizer, eval mode
# --------------------------------------------------------------------------------------

def leaky(x):
    return F.leaky_relu(x, 0.1)                   # vq.py:94-95 (slope always 0.1, quirk Q10)


def resblock(x, P, groups=16):
    """ResBlock vq.py:228-242."""
    h = F.conv2d(x, P['net.0.weight'], P['net.0.bias'], padding=1)
    h = leaky(F.group_norm(h, groups, P['net.1.weight'], P['net.1.bias']))
    h = F.conv2d(h, P['net.3.weight'], P['net.3.bias'], padding=1)
    h = leaky(F.group_norm(h, groups, P['net.4.weight'], P['net.4.bias']))
    h = F.conv2d(h, P['net.6.weight'], P['net.6.bias'])
    return h + x


def glu_resblock(x, P, groups=16):
    """GLUResBlock vq.py:212-226."""
    h = F.glu(F.conv2d(x, P['net.0.weight'], P['net.0.bias'], padding=1), dim=1)
    h = F.group_norm(h, groups, P['net.2.weight'], P['net.2.bias'])
    h = F.glu(F.conv2d(h, P['net.3.weight'], P['net.3.bias'], padding=1), dim=1)
    h = F.group_norm(h, groups, P['net.5.weight'], P['net.5.bias'])
    h = F.conv2d(h, P['net.6.weight'], P['net.6.bias'])
    return h + x


def layernorm_chan(x, g, b, eps=1e-5):
    """LayerNormChan vq.py:129-143."""
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) / (var + eps).sqrt() * g + b


def cpb_bias(P, fmap):
    """ContinuousPositionBias vq.py:178-210 -> (heads, n, n) bias."""
    pos = torch.arange(fmap)
    grid = torch.stack(torch.meshgrid(pos, pos, indexing='ij')).reshape(2, -1).t()
    rel = grid[:, None, :] - grid[None, :, :]
    rel = torch.sign(rel) * torch.log(rel.abs() + 1)
    h = rel.float()
    i = 0
    while f'net.{i}.0.weight' in P:
        h = leaky(h @ P[f'net.{i}.0.weight'].t() + P[f'net.{i}.0.bias'])
        i += 1
    h = h @ P[f'net.{i}.weight'].t() + P[f'net.{i}.bias']
    return h.permute(2, 0, 1)


def vqgan_attention(x, P, heads=8):
    """VQGanAttention vq.py:244-286 incl. quirk Q9 (l2norm over the SPATIAL axis)."""
    B, C, Hh, Ww = x.shape
    qkv = F.conv2d(x, P['to_qkv.weight'])
    q, k, v = qkv.chunk(3, dim=1)
    rs = lambda t: t.reshape(B, heads, -1, Hh * Ww)          # b h c (x y)
    q, k, v = rs(q), rs(k), rs(v)
    q, k = F.normalize(q, dim=-1), F.normalize(k, dim=-1)
    sim = torch.einsum('bhci,bhcj->bhij', q, k) * P['scale'].exp()
    sim = sim + cpb_bias(sub(P, 'cpb'), Hh)
    alpha = 32 ** 2                                          # stable_softmax vq.py:97-100
    t = sim / alpha
    t = t - t.amax(dim=-1, keepdim=True)
    attn = (t * alpha).softmax(dim=-1)
    out = torch.einsum('bhij,bhcj->bhci', attn, v).reshape(B, -1, Hh, Ww)
    out = F.conv2d(out, P['to_out.weight'], P['to_out.bias'])
    return layernorm_chan(out, P['post_norm.g'], P['post_norm.b']) + x


def vae_encode_fmap(img, P, num_layers, num_resnet_blocks=1, use_attn=True, groups=16, heads=8):
    """the `encoders` ModuleList of VQGanVAE (built vq.py:351-365, run vq.py:432-433):
    encoders.0 = Conv(c->dim, 5, pad 2); then per layer Sequential(Conv 4 s2 p1, LeakyReLU);
    last layer followed by ResBlock x n and VQGanAttention."""
    E = sub(P, 'encoders')
    h = F.conv2d(img, E['0.weight'], E['0.bias'], padding=E['0.weight'].shape[-1] // 2)
    i = 1
    for layer in range(num_layers):
        h = leaky(F.conv2d(h, E[f'{i}.0.weight'], E[f'{i}.0.bias'], stride=2, padding=1))
        i += 1
        if layer == num_layers - 1:
            for _ in range(num_resnet_blocks):
                h = resblock(h, sub(E, str(i)), groups)
                i += 1
            if use_attn:
                h = vqgan_attention(h, sub(E, str(i)), heads)
                i += 1
    return h


def vq_eval_lookup(fmap, codebook, project_in_w=None, project_in_b=None):
    """PARITY UNPINNED (third-party vector_quantize_pytorch, not available here; call sites
    vq.py:368-378, 435).  Eval path, use_cosine_sim=True: x = project_in(b (h w) c);
    idx = argmax_c( l2norm(x) . l2norm(codebook)^T ), lowest index on ties.
    Returns (indices (B,h,w) int64, sim top-2 gap (B,h,w))."""
    B, C, Hh, Ww = fmap.shape
    x = fmap.permute(0, 2, 3, 1).reshape(B, Hh * Ww, C)
    if project_in_w is not None:
        x = x @ project_in_w.t() + (project_in_b if project_in_b is not None else 0)
    xn = F.normalize(x, dim=-1)
    cn = F.normalize(codebook, dim=-1)
    sim = xn @ cn.t()
    top2 = sim.topk(2, dim=-1).values
    idx = sim.argmax(dim=-1)
    return idx.reshape(B, Hh, Ww), (top2[..., 0] - top2[..., 1]).reshape(B, Hh, Ww)


def get_video_indices(video, P, num_layers, **kw):
    """VQGanVAE.get_video_indices vq.py:452-458. P = state dict of the VAE; expects the VQ
    parameters under the names used by nuwa_pytorch_amd's VectorQuantize restatement:
    vq.project_in.{weight,bias}, vq.codebook (C, codebook_dim)."""
    b, f = video.shape[:2]
    fm = vae_encode_fmap(video.reshape(b * f, *video.shape[2:]), P, num_layers, **kw)
    idx, gap = vq_eval_lookup(fm, P['vq.codebook'], P.get('vq.project_in.weight'), P.get('vq.project_in.bias'))
    return idx.reshape(b, f, *idx.shape[1:]), gap.reshape(b, f, *gap.shape[1:])
