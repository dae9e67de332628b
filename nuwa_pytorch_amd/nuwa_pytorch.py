"""Host-side mirror of the reference's nuwa_pytorch/nuwa_pytorch.py (np.py) for the video-decoder
training path: same class names, constructor kwargs, forward()/generate() signatures and
state_dict keys, with every hot forward/backward routed to libamdnuwa (gfx950 HIP kernels) through
`ops.py`.  nn.Linear / nn.Conv2d / nn.LayerNorm / nn.Embedding sub-modules are kept purely as
PARAMETER CONTAINERS (identical keys, shapes and default initialisation as the reference); their
own forward() is never called on the decoder path.

Scope: decoder stack, embeddings, logits/loss = HIP.  The text encoder (row f1,
256 tokens once per step) runs its attention / FeedForward blocks through the same kernels.  NUWAVideoAudio (cfg 5) lives in
video_audio.py; NUWASketch (row f4) shares the decoder kernels, with SparseCross2DNA and the sketch encoder on PyTorch-ROCm ops.
"""
import functools
from functools import partial

import torch
import torch.nn.functional as F
from torch import nn, einsum

from . import kernels as K
from . import ops

MList = nn.ModuleList


def exists(val):
    return val is not None


def default(val, d):
    return val if exists(val) else d


def cast_tuple(val, size=1):
    return val if isinstance(val, tuple) else (val,) * size


def calc_same_padding(kernel_size, dilation=1):
    return dilation * (kernel_size - 1) // 2


def mult_reduce(arr):
    return functools.reduce(lambda x, y: x * y, arr, 1)


def eval_decorator(fn):
    def inner(model, *args, **kwargs):
        was_training = model.training
        model.eval()
        out = fn(model, *args, **kwargs)
        model.train(was_training)
        return out
    return inner


def sample_top_fraction(logits, filter_thres=0.9, temperature=1.):
    """one token id per row of `logits` (b, C): restrict to the top max(1, int((1 - filter_thres) C)) logits, then a Gumbel-max draw at
    `temperature` -- the distribution of the reference's top_k + gumbel_sample pair (np.py:55-65, 1713-1720); the noise is drawn for
    the kept logits only (the filtered ones sit at -inf there and can never win)"""
    keep = max(int((1 - filter_thres) * logits.shape[-1]), 1)
    vals, ids = logits.topk(keep, dim=-1)
    if keep == 1:
        return ids[..., 0]
    u = torch.rand_like(vals).clamp_(min=1e-20)
    gumbel = -torch.log((-torch.log(u)).clamp_(min=1e-20))
    return ids.gather(-1, (vals / temperature + gumbel).argmax(dim=-1, keepdim=True))[..., 0]


def bernoulli_rows(count, prob, device):
    """bool (count,): True with probability `prob` (classifier-free-guidance condition dropout, np.py:67-68, 1952-1954)"""
    return torch.rand(count, device=device) < prob


def map_in_chunks(t, fn, chunks=10):
    """fn over at most `chunks` slices of the leading axis, results concatenated (bounds the VAE decoder's batch, np.py:70-72)"""
    return torch.cat([fn(piece) for piece in t.chunk(chunks, dim=0)], dim=0)


def lookback_window(ids, tokens_per_frame, max_frames):
    """the suffix of a generated id sequence that still fits the decoder's video shape (np.py:1876-1881): whole frames plus the
    frame being written"""
    n = ids.shape[1]
    if n <= tokens_per_frame * max_frames:
        return ids
    partial = n % tokens_per_frame
    keep = (max_frames - (1 if partial else 0)) * tokens_per_frame + partial
    return ids[:, -keep:]


def causal_neighbor_mask(video_shape, kernel_size, dilation):
    """(N, K+1) bool, True = tap falls in the causal zero padding; column 0 (<bos>) never masked.
    Same content as the reference's `mask` buffer (np.py:442-457), computed by index arithmetic."""
    Fr, H, W = video_shape
    kf, kh, kw = kernel_size
    df, dh, dw = dilation
    f = torch.arange(Fr)[:, None, None]
    y = torch.arange(H)[None, :, None]
    w = torch.arange(W)[None, None, :]
    cols = [torch.zeros(Fr * H * W, dtype=torch.bool)]
    for a in range(kf):
        for b in range(kh):
            for c in range(kw):
                bad = ((f - (kf - 1 - a) * df) < 0) | ((y - (kh - 1 - b) * dh) < 0) | ((w - (kw - 1 - c) * dw) < 0)
                cols.append(bad.expand(Fr, H, W).reshape(-1))
    return torch.stack(cols, dim=1)


def neighbor_positions(video_shape, kernel_size, dilation, causal):
    """(N, K) int64: raster position of tap slot (a, b, c) of every query position, -1 where the tap falls in the zero padding.
    causal: all padding on the low side (np.py:427); otherwise symmetric 'same' padding (np.py:429)."""
    Fr, H, W = video_shape
    kf, kh, kw = kernel_size
    df, dh, dw = dilation
    f = torch.arange(Fr)[:, None, None]
    y = torch.arange(H)[None, :, None]
    w = torch.arange(W)[None, None, :]
    cols = []
    for a in range(kf):
        for b in range(kh):
            for c in range(kw):
                if causal:
                    ff, yy, ww = f - (kf - 1 - a) * df, y - (kh - 1 - b) * dh, w - (kw - 1 - c) * dw
                else:
                    ff, yy, ww = f + (a - (kf - 1) // 2) * df, y + (b - (kh - 1) // 2) * dh, w + (c - (kw - 1) // 2) * dw
                ok = (ff >= 0) & (ff < Fr) & (yy >= 0) & (yy < H) & (ww >= 0) & (ww < W)
                pos = ((ff * H + yy) * W + ww).expand(Fr, H, W)
                cols.append(torch.where(ok.expand(Fr, H, W), pos, torch.full_like(pos, -1)).reshape(-1))
    return torch.stack(cols, dim=1)


# ---------------------------------------------------------------------------------------------------
# normalisations
# ---------------------------------------------------------------------------------------------------

class StableLayerNorm(nn.Module):
    """np.py:88-95"""

    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return ops.StableLNFn.apply(x, self.norm.weight, self.norm.bias)


class SandwichNorm(nn.Module):
    """np.py:112-128.  Standalone call = three nodes (LN, fn, LN); inside Transformer the whole
    `SandwichNorm(fn)(x) + x` is one fused node (see Transformer.forward)."""

    def __init__(self, *, dim, fn):
        super().__init__()
        self.prenorm = nn.LayerNorm(dim)
        self.postnorm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        x = ops.LayerNormFn.apply(x, self.prenorm.weight, self.prenorm.bias, False)
        x = self.fn(x, **kwargs)
        return ops.LayerNormFn.apply(x, self.postnorm.weight, self.postnorm.bias, False)

    # -- fused path -------------------------------------------------------------------------------
    def _inner(self, context=None, seq_len=None):
        """(inner module, shift or None) when fn is one of the fusable hot modules, else None.
        An Attention is fusable only as cross-attention (context given)."""
        fn, shift = self.fn, None
        if isinstance(fn, ShiftVideoTokens):
            if fn.shift_time:
                return None
            if fn.shift_space:
                shift = fn.image_size
            fn = fn.fn
        elif isinstance(fn, ShiftAudioTokens):            # cfg 5 audio tower: the one-row channel shift rides in the pre-norm's store
            shift, fn = -1, fn.fn
            if isinstance(fn, SparseCausal2DNA) and fn._hip_ok():
                return fn, shift
        if (isinstance(fn, FeedForward) and not fn._dropout_active()) or (isinstance(fn, Sparse3DNA) and fn._hip_ok()):
            return fn, shift
        if isinstance(fn, Attention) and context is not None and fn._hip_ok(context.shape[1]):
            return fn, shift
        if isinstance(fn, Attention) and context is None and seq_len is not None and fn._hip_ok(seq_len):
            return fn, shift                      # non-causal self-attention (text encoder): same kernels, keys = the query rows
        if isinstance(fn, SparseCross2DNA) and context is not None and fn._hip_ok(context.shape[1]):
            return fn, shift                      # NUWASketch decoder: 2-D nearby cross-attention on the 3DNA kernels (row f4)
        return None

    def fused_residual(self, x, resid=None, context=None, context_mask=None, mask=None, rotary_pos_emb=None, chain=None, minus=False):
        """x_out = (resid if given else x) + postnorm(fn(prenorm(x)))  as one autograd node.
        minus (with resid; the reversible reconstruction): the VALUE is resid - postnorm(...), the backward stays that of the sum -- the
        node then returns a reversible block's input (x2 = y2 - g(y1)) and, fed the block's output gradient, g's gradients.
        chain = (handoff_in or None, next SandwichNorm or None, next block's fmap, handoff_out dict): lets this block's post-norm
        kernel also emit the next block's pre-norm output (see ops.SandwichBlockFn)"""
        B, n, D = x.shape
        inner, fmap = self._inner(context, seq_len=n)
        meta = inner._meta(B, n, x.device, context=context, context_mask=context_mask, mask=mask, rotary_pos_emb=rotary_pos_emb)
        if fmap is not None:
            if D % 32:
                raise RuntimeError('fused token shift needs dim % 32 == 0')
            meta['shift'] = (n, fmap)
        if minus:
            assert resid is not None and chain is None
            meta['resid_minus'] = True
        if chain is not None:
            hin, nxt, nxt_fmap, hout = chain[:4]
            nxt_ctx = chain[4].get('context') if len(chain) > 4 else None        # the next block's text context (cross-attention)
            meta['handoff_in'] = hin
            if nxt is not None and not (nxt_fmap is not None and D % 32):
                nxt_fn = nxt.fn.fn if isinstance(nxt.fn, (ShiftVideoTokens,)) else nxt.fn
                nxt_kind = None                  # what the next block's first GEMM is (it may want an fp16 copy of its input: ops.SandwichBlockFn)
                if isinstance(nxt_fn, FeedForward):
                    nxt_kind = ('ff', nxt_fn.net[3].weight.shape[1], (nxt_fn.net[0].weight, nxt_fn.net[3].weight))
                elif isinstance(nxt_fn, Sparse3DNA) and nxt_fn.causal:
                    nxt_kind = ('s3', nxt_fn.to_q.weight.shape[0], nxt_fn._meta(B, n, x.device)['geom'], (nxt_fn.to_q.weight, nxt_fn.to_kv.weight),
                                nxt_fn.to_out.weight, nxt_fn.rel_pos_bias is not None)
                elif isinstance(nxt_fn, Attention) and nxt_ctx is not None and nxt_fn._hip_ok(nxt_ctx.shape[1]):
                    nxt_kind = ('x', nxt_fn.to_q.weight.shape[0], K.x_geom(B, n, nxt_ctx.shape[1], nxt_fn.heads, nxt_fn.dim_head), {},
                                (nxt_fn.to_q.weight, nxt_fn.to_out.weight))
                meta['next_pre'] = (nxt.prenorm.weight, nxt.prenorm.bias, (n, nxt_fmap) if nxt_fmap is not None else None, nxt_kind)
                meta['handoff_out'] = hout
        return ops.SandwichBlockFn.apply(x, resid, context if isinstance(inner, (Attention, SparseCross2DNA)) else None, meta,
                                         self.prenorm.weight, self.prenorm.bias, self.postnorm.weight,
                                         self.postnorm.bias, *inner._params())


# ---------------------------------------------------------------------------------------------------
# rotary (text encoder only)
# ---------------------------------------------------------------------------------------------------

class RotaryEmbedding(nn.Module):
    def __init__(self, dim):
        super().__init__()
        inv_freq = 1. / (10000 ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer('inv_freq', inv_freq)

    def forward(self, seq_len, device):
        t = torch.arange(seq_len, device=device).type_as(self.inv_freq)
        freqs = torch.einsum('i , j -> i j', t, self.inv_freq)
        return torch.cat((freqs, freqs), dim=-1)


def rotate_half(x):
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary_pos_emb(freqs, t):
    rot_dim = freqs.shape[-1]
    t, t_pass = t[..., :rot_dim], t[..., rot_dim:]
    t = (t * freqs.cos()) + (rotate_half(t) * freqs.sin())
    return torch.cat((t, t_pass), dim=-1)


# ---------------------------------------------------------------------------------------------------
# token shift
# ---------------------------------------------------------------------------------------------------

class ShiftVideoTokens(nn.Module):
    """np.py:185-253.  Inside the fused decoder block the shift costs nothing (it is an address
    offset in the consumer GEMM's loader, see csrc/gemm.hip); this standalone forward materialises
    it with index ops for arbitrary wrapped `fn`."""

    def __init__(self, fn, image_size, shift_space=True, shift_time=False):
        super().__init__()
        self.fn = fn
        self.image_size = image_size
        self.shift_time = shift_time
        self.shift_space = shift_space

    def forward(self, x, **kwargs):
        if not self.shift_time and not self.shift_space:
            return self.fn(x, **kwargs)
        fmap = self.image_size
        b, n, D = x.shape
        if n == 1:
            return self.fn(x, **kwargs)
        nchunk = 5 if (self.shift_space and self.shift_time) else (4 if self.shift_space else 3)
        cs = -(-D // nchunk)
        p = torch.arange(n - 1, device=x.device)
        yy, ww, ff = (p // fmap) % fmap, p % fmap, p // (fmap * fmap)
        xv = x[:, 1:]
        parts, c0 = [], 0
        plan = ([('f', fmap * fmap, ff)] if self.shift_time else []) + \
               ([('h', fmap, yy), ('w', 1, ww)] if self.shift_space else [])
        for _, off, coord in plan:
            c1 = min(c0 + cs, D)
            src = (p - off).clamp(min=0)
            parts.append(xv[:, src, c0:c1] * (coord > 0).to(x.dtype)[None, :, None])
            c0 = c1
        parts.append(xv[:, :, c0:])
        x = torch.cat((x[:, :1], torch.cat(parts, dim=-1)), dim=1)
        return self.fn(x, **kwargs)


# ---------------------------------------------------------------------------------------------------
# feed forward
# ---------------------------------------------------------------------------------------------------

class GEGLU(nn.Module):
    def forward(self, x):
        x, gate = x.chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    """np.py:260-286; forward = GEMM -> fused GEGLU kernel -> GEMM (chunk_size only limits memory in the
    reference and is numerically a no-op: kept in the signature, ignored)."""

    def __init__(self, *, dim, mult=4, dropout=0., chunk_size=None):
        super().__init__()
        inner_dim = (dim * mult * 2) // 3
        self.chunk_size = chunk_size
        self.net = nn.Sequential(
            nn.Linear(dim, inner_dim * 2, bias=False),
            GEGLU(),
            nn.Dropout(dropout),
            nn.Linear(inner_dim, dim, bias=False)
        )
        self._cache = ops.WeightCache()

    def _params(self):
        return (self.net[0].weight, self.net[3].weight)

    def _dropout_active(self):
        return self.training and self.net[2].p > 0

    def _meta(self, B, n, device, **_):
        assert not self._dropout_active()
        return dict(kind='ff', cache=self._cache)

    def forward(self, x):
        B, n, D = x.shape
        if self._dropout_active() and x.is_cuda:
            # ff_dropout > 0 in training (np.py:276): the GEGLU output passes through nn.Dropout.  Both products, the gate and its backward stay on
            # libamdnuwa (standalone node, bf16 backward; the mask comes from torch's RNG stream and is kept for the backward); the block is not
            # part of the fused / chained stack (SandwichNorm._inner).  Inside a reversible stack the forward runs twice per step and there is no
            # RNG replay here (Deterministic): those modules keep the torch-op forward.  Every BASELINE config trains with dropout 0.
            if getattr(self, '_no_hip_dropout', False):
                return self.net(x)
            meta = dict(kind='ff', cache=self._cache, drop_p=float(self.net[2].p))
            return ops.InnerFn.apply(x, None, meta, *self._params())
        return ops.InnerFn.apply(x, None, self._meta(B, n, x.device), *self._params())


# ---------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------

class Attention(nn.Module):
    """np.py:290-379.  With `context` (the decoder's text cross-attention) and as non-causal self-attention (the text encoder, row f1:
    keys / values = the query rows, rotary on q, k and v) the core runs on the MFMA cross-attention kernels (xattn6; xattn2 / xattn for
    the shapes it does not take).  Outside `_hip_ok` -- causal=True, more than 287 keys, more than 8 heads, dim_head not 32 / 64,
    attention dropout > 0 in training -- the forward below runs on PyTorch-ROCm ops."""

    def __init__(self, *, dim, heads=8, dim_head=64, causal=False, dropout=0.):
        super().__init__()
        inner_dim = heads * dim_head
        self.heads = heads
        self.dim_head = dim_head
        self.causal = causal
        self.scale = dim_head ** -0.5
        self.null_k = nn.Parameter(torch.randn(heads, 1, dim_head))
        self.null_v = nn.Parameter(torch.randn(heads, 1, dim_head))
        self.talking_heads = nn.Conv2d(heads, heads, 1, bias=False)
        self.dropout = nn.Dropout(dropout)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)
        self._cache = ops.WeightCache()

    def _params(self):
        return (self.null_k, self.null_v, self.talking_heads.weight, self.to_q.weight, self.to_kv.weight, self.to_out.weight)

    def _hip_ok(self, n_keys):
        """geometry the cross-attention kernels cover (they also serve non-causal SELF-attention: keys / values = the query rows)"""
        return (not self.causal) and self.dim_head in (32, 64) and self.heads <= 8 and n_keys + 1 <= 288 and \
            not (self.training and self.dropout.p > 0)

    def _meta(self, B, n, device, context=None, context_mask=None, mask=None, rotary_pos_emb=None, **_):
        assert not (self.training and self.dropout.p > 0)      # (_hip_ok routes attn_dropout > 0 in training to the torch-op forward)
        self_kv = context is None
        T = n if self_kv else context.shape[1]
        key_mask = mask if self_kv else context_mask
        g = K.x_geom(B, n, T, self.heads, self.dim_head)
        mask_u8 = key_mask.to(torch.uint8).contiguous() if exists(key_mask) else None
        meta = dict(kind='xattn', cache=self._cache, xgeom=g, mask_u8=mask_u8, save=torch.is_grad_enabled())
        if self_kv:                                   # text encoder (row f1): rotary on q, k and v (quirk Q11), key mask = `mask`
            meta['self_kv'] = True
            if exists(rotary_pos_emb):
                meta['rotary'] = rotary_pos_emb.detach().float()
        return meta

    def forward(self, x, mask=None, context=None, context_mask=None, rotary_pos_emb=None):
        B, n, _ = x.shape
        if x.is_cuda and self._hip_ok(context.shape[1] if exists(context) else n):
            meta = self._meta(B, n, x.device, context, context_mask, mask=mask, rotary_pos_emb=rotary_pos_emb)
            return ops.InnerFn.apply(x, context, meta, *self._params())
        return self._forward_torch(x, mask=mask, context=context, context_mask=context_mask, rotary_pos_emb=rotary_pos_emb)

    def _forward_torch(self, x, mask=None, context=None, context_mask=None, rotary_pos_emb=None):
        """np.py:315-379 on PyTorch-ROCm ops, for what the kernels do not cover: causal self-attention, other head sizes, and
        attn_dropout > 0 in training.  Keys = [null key | context or x]; one additive bias carries the key mask and the causal band."""
        b, n, h, dh = x.shape[0], x.shape[1], self.heads, self.dim_head
        src = x if context is None else context
        heads_of = lambda t: t.reshape(b, t.shape[1], -1, dh).transpose(1, 2)               # (b, heads, len, dh)
        q = heads_of(self.to_q(x))
        k, v = (heads_of(t) for t in self.to_kv(src).chunk(2, dim=-1))
        if context is None and rotary_pos_emb is not None:
            q, k, v = (apply_rotary_pos_emb(rotary_pos_emb, t) for t in (q, k, v))         # v too (quirk Q11)
        k = torch.cat((self.null_k.expand(b, -1, -1, -1), k), dim=2)
        v = torch.cat((self.null_v.expand(b, -1, -1, -1), v), dim=2)
        scores = torch.matmul(q * self.scale, k.transpose(-1, -2))
        j = k.shape[2]
        hidden = torch.zeros(b, 1, n if self.causal else 1, j, dtype=torch.bool, device=x.device)
        key_mask = mask if context is None else context_mask
        if key_mask is not None:
            hidden = hidden | ~F.pad(key_mask, (1, 0), value=True)[:, None, None, :]
        if self.causal:                                        # query i sees the null key and keys 0..i of the sequence
            hidden = hidden | (torch.arange(j, device=x.device)[None, :] > torch.arange(n, device=x.device)[:, None] + (j - n))
        scores = scores.masked_fill(hidden, -torch.finfo(scores.dtype).max)
        probs = self.dropout(self.talking_heads(scores.softmax(dim=-1, dtype=torch.float32)))
        out = torch.matmul(probs, v).transpose(1, 2).reshape(b, n, h * dh)
        return self.to_out(out)


class Sparse3DNA(nn.Module):
    """np.py:381-613.  forward = qkv GEMM -> fused gfx950 3DNA kernel (no unfold materialisation) -> to_out GEMM, for the causal
    window of the video decoder AND the symmetric window of NUWASketch's sketch encoder (causal=False, np.py:429: the same kernels
    with the tap origin in the middle of the window).  `query_num_frames_chunk` only bounds the reference's unfolded tensors; it is
    accepted and ignored (chunked == unchunked bit for bit in the reference)."""

    def __init__(self, dim, video_shape, kernel_size=3, dilation=1, heads=8, dim_head=64, dropout=0., causal=False,
                 query_num_frames_chunk=None, rel_pos_bias=False):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.causal = causal
        self.dropout = nn.Dropout(dropout)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=False)
        self.talking_heads = nn.Conv2d(heads, heads, 1, bias=False)
        self.to_out = nn.Linear(inner_dim, dim)
        self.dilation = cast_tuple(dilation, size=3)
        self.kernel_size = cast_tuple(kernel_size, size=3)
        assert all(map(lambda n: n % 2 == 1, self.kernel_size)), 'kernel size must be odd'
        self.kernel_numel = mult_reduce(self.kernel_size)
        # np.py:416: one bias per (key tap, head), axial over the kernel (the reference's add only broadcasts for batch 1; here
        # the same per-head bias is applied to every sample)
        self.rel_pos_bias = AxialPositionalEmbedding(heads, shape=self.kernel_size) if rel_pos_bias else None
        self.video_shape = video_shape
        max_frames, fmap_size, _ = video_shape
        self.max_num_tokens = max_frames * fmap_size * fmap_size
        self.query_num_frames_chunk = default(query_num_frames_chunk, max_frames)
        if causal:
            self.register_buffer('mask', causal_neighbor_mask(video_shape, self.kernel_size, self.dilation))
        else:
            # NUWASketch's sketch encoder (sketch_enc_use_sparse_3dna=True): symmetric window (row f4); the neighbour table is
            # only used by the PyTorch-op formulation kept for geometries the kernels do not take
            nbr = neighbor_positions(video_shape, self.kernel_size, self.dilation, causal=False)
            self.register_buffer('mask', F.pad(nbr < 0, (1, 0), value=False))
            self.register_buffer('_nbr', nbr, persistent=False)
        self._cache = ops.WeightCache()

    def _params(self):
        p = (self.to_q.weight, self.to_kv.weight, self.talking_heads.weight, self.to_out.weight, self.to_out.bias)
        if self.rel_pos_bias is not None:
            taps = self.rel_pos_bias().reshape(self.kernel_numel, self.heads)             # [taps, heads], raster (a, b, c) order
            p = p + (torch.cat((taps.new_zeros(1, self.heads), taps), 0).float(),)       # key slot 0 = <bos>: no bias
        return p

    def _meta(self, B, n, device, **_):
        if self.training and self.dropout.p > 0:
            raise NotImplementedError('dropout inside Sparse3DNA is not supported (never enabled by Transformer, quirk Q14)')
        assert n - 1 <= self.max_num_tokens, 'sequence longer than the video shape allows'
        g = K.s3_geom(B, n, self.video_shape, self.kernel_size, self.dilation, self.heads, self.dim_head, causal=self.causal)
        return dict(kind='s3', cache=self._cache, geom=g)

    def _hip_ok(self):
        """geometry the 3DNA kernels take.  The causal decoder path never falls back (an unsupported shape raises from the library);
        the symmetric variant keeps its PyTorch-op formulation for head sizes / widths outside the kernels"""
        if self.causal:
            return True
        if self.training and self.dropout.p > 0:
            return False
        # head size / width limits AND the LDS the window's key-slot tables need (a 7x7x7 sketch-encoder window does not fit)
        return K.s3_supported(self.video_shape, self.kernel_size, self.dilation, self.heads, self.dim_head, causal=False)

    def forward(self, x, **kwargs):
        B, n, _ = x.shape
        if not self.causal and not (x.is_cuda and self._hip_ok()):
            return self._forward_noncausal(x)
        return ops.InnerFn.apply(x, None, self._meta(B, n, x.device), *self._params())

    def _forward_noncausal(self, x):
        """np.py:459-613 with causal=False, as a gather over the neighbour table instead of unfoldNd.  The reference treats
        row 0 as <bos> here too, pads the remaining rows to whole frames with zero rows, and masks only the taps that leave the
        FULL video shape -- so zero rows (and frames the sequence does not reach) are attended with score 0."""
        b, n, _ = x.shape
        h = self.heads
        q, (k, v) = self.to_q(x), self.to_kv(x).chunk(2, dim=-1)
        if n == 1:
            return self.to_out(v)
        q, k, v = (t.reshape(b, n, h, -1).transpose(1, 2) for t in (q, k, v))                 # b h n d
        nq = n - 1
        qs = q[:, :, 1:] * self.scale
        tab = self._nbr[:nq]                                                                    # (nq, K)
        valid = tab >= 0
        idx = torch.where(valid & (tab < nq), tab, torch.full_like(tab, nq)).reshape(-1)        # row nq = the zero row
        zero = k.new_zeros(b, h, 1, k.shape[-1])
        kg = torch.cat((k[:, :, 1:], zero), 2)[:, :, idx].reshape(b, h, nq, -1, k.shape[-1])
        vg = torch.cat((v[:, :, 1:], zero), 2)[:, :, idx].reshape(b, h, nq, -1, v.shape[-1])
        sim = torch.cat((einsum('b h i d, b h d -> b h i', qs, k[:, :, 0])[..., None],
                         einsum('b h i d, b h i j d -> b h i j', qs, kg)), dim=-1)
        if self.rel_pos_bias is not None:
            sim = sim + F.pad(self.rel_pos_bias().reshape(self.kernel_numel, h).t(), (1, 0))[None, :, None, :]
        sim = sim.masked_fill(~F.pad(valid, (1, 0), value=True)[None, None], -torch.finfo(sim.dtype).max)
        attn = sim.softmax(dim=-1, dtype=torch.float32)
        attn = self.dropout(einsum('g h, b h i j -> b g i j', self.talking_heads.weight.reshape(h, h), attn))
        out = attn[..., :1] * v[:, :, :1] + einsum('b h i j, b h i j d -> b h i d', attn[..., 1:], vg)
        out = torch.cat((v[:, :, :1], out), dim=2)
        return self.to_out(out.transpose(1, 2).reshape(b, n, -1))


class SparseCross2DNA(nn.Module):
    """np.py:761-901: cross-attention of video tokens to the sketch tokens in a 2-D neighbourhood of the same feature-map
    position in EVERY sketch frame (+ a learned null key); the <bos> query attends to all sketch tokens, without talking heads.
    On the MI355X the windowed queries run on the 3DNA kernels pointed at the sketch context (amdnuwa_cross2dna_*, row f4; the B
    <bos> rows are glue arithmetic between the kernels); `_forward_torch` keeps the gather formulation on PyTorch ops for shapes the
    kernels do not take (other head sizes, attention dropout in training)."""

    def __init__(self, *, dim, image_size, heads=8, dim_head=64, dropout=0., kernel_size=3, dilation=1):
        super().__init__()
        inner_dim = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.null_k = nn.Parameter(torch.randn(heads, 1, dim_head))
        self.null_v = nn.Parameter(torch.randn(heads, 1, dim_head))
        self.talking_heads = nn.Conv3d(heads, heads, 1, bias=False)
        self.dropout = nn.Dropout(dropout)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)
        self.image_size, self.kernel_size, self.dilation = image_size, kernel_size, dilation
        self.padding = calc_same_padding(kernel_size, dilation)
        nbr = neighbor_positions((1, image_size, image_size), (1, kernel_size, kernel_size), (1, dilation, dilation), causal=False)
        self.register_buffer('_nbr', nbr, persistent=False)                                    # (fmap^2, k^2), -1 = padding
        self.dim_head = dim_head
        self._cache = ops.WeightCache()

    def _params(self):
        return (self.null_k, self.null_v, self.talking_heads.weight, self.to_q.weight, self.to_kv.weight, self.to_out.weight)

    def _hip_ok(self, ctx_len):
        tpf = self.image_size ** 2
        if ctx_len <= 0 or ctx_len % tpf or (self.training and self.dropout.p > 0):
            return False
        # the window holds kernel^2 slots of EVERY sketch frame: with many frames / a large kernel its LDS tables outgrow the CU
        # (e.g. kernel 5 with 6 frames, kernel 3 with 16) and the PyTorch-op formulation below takes over
        return K.s3_supported((1, self.image_size, self.image_size), (ctx_len // tpf, self.kernel_size, self.kernel_size),
                              (1, self.dilation, self.dilation), self.heads, self.dim_head, causal=False)

    def _meta(self, B, n, device, context=None, context_mask=None, **_):
        tpf = self.image_size ** 2
        T = context.shape[1]
        frames = max(1, -(-(n - 1) // tpf))
        g = K.s3_geom(B, n, (frames, self.image_size, self.image_size), (T // tpf, self.kernel_size, self.kernel_size),
                      (1, self.dilation, self.dilation), self.heads, self.dim_head, causal=False)
        mask_u8 = context_mask.to(torch.uint8).contiguous() if exists(context_mask) else None
        return dict(kind='xc2', cache=self._cache, geom=g, ctx_T=T, mask_u8=mask_u8)

    def forward(self, x, *, context, context_mask=None, **kwargs):
        if x.is_cuda and self._hip_ok(context.shape[1]):
            B, n, _ = x.shape
            return ops.InnerFn.apply(x, context, self._meta(B, n, x.device, context=context, context_mask=context_mask), *self._params())
        return self._forward_torch(x, context=context, context_mask=context_mask)

    def _forward_torch(self, x, *, context, context_mask=None, **kwargs):
        b, n, h, device = x.shape[0], x.shape[1], self.heads, x.device
        tpf = self.image_size ** 2
        kn = self.kernel_size ** 2
        if not exists(context_mask):
            context_mask = torch.ones((b, context.shape[-2]), dtype=torch.bool, device=device)
        mask_value = -torch.finfo(x.dtype).max
        q, (k, v) = self.to_q(x), self.to_kv(context).chunk(2, dim=-1)
        q, k, v = (t.reshape(b, t.shape[1], h, -1).transpose(1, 2) for t in (q, k, v))        # b h n d
        q = q * self.scale
        q_bos, q = q[:, :, 0], q[:, :, 1:]
        nk, nv = self.null_k[None].expand(b, -1, -1, -1), self.null_v[None].expand(b, -1, -1, -1)
        sim_bos = einsum('b h d, b h j d -> b h j', q_bos, torch.cat((nk, k), dim=-2))
        sim_bos = sim_bos.masked_fill(~F.pad(context_mask[:, None], (1, 0), value=True), mask_value)
        out_bos = einsum('b h j, b h j d -> b h d', sim_bos.softmax(dim=-1, dtype=torch.float32), torch.cat((nv, v), dim=-2))
        out_bos = out_bos.reshape(b, 1, -1)
        if n == 1:
            return self.to_out(out_bos)
        f = context.shape[-2] // tpf
        tab = self._nbr                                                                         # (tpf, kn)
        valid = tab >= 0
        idx = tab.clamp(min=0).reshape(-1)
        # keys / values of position i: for every sketch frame the kn taps around i -> slot order (frame, tap), np.py:855
        kf = k.reshape(b, h, f, tpf, -1)[:, :, :, idx].reshape(b, h, f, tpf, kn, -1).permute(0, 1, 3, 2, 4, 5).reshape(b, h, tpf, f * kn, -1)
        vf = v.reshape(b, h, f, tpf, -1)[:, :, :, idx].reshape(b, h, f, tpf, kn, -1).permute(0, 1, 3, 2, 4, 5).reshape(b, h, tpf, f * kn, -1)
        vmask = valid[None, None, :, None, :, None].to(kf.dtype).expand(1, 1, tpf, f, kn, 1).reshape(1, 1, tpf, f * kn, 1)
        kf, vf = kf * vmask, vf * vmask                                                        # F.unfold zero-pads
        kf = torch.cat((nk[:, :, None].expand(-1, -1, tpf, -1, -1), kf), dim=-2)
        vf = torch.cat((nv[:, :, None].expand(-1, -1, tpf, -1, -1), vf), dim=-2)
        nq = q.shape[-2]
        qpad = (-nq) % tpf
        qf = F.pad(q, (0, 0, 0, qpad)).reshape(b, h, -1, tpf, q.shape[-1])                     # b h F i d
        sim = einsum('b h f i d, b h i j d -> b h f i j', qf, kf)
        cm = context_mask.reshape(b, f, tpf)[:, :, idx].reshape(b, f, tpf, kn) & valid[None, None]
        cm = F.pad(cm.permute(0, 2, 1, 3).reshape(b, 1, 1, tpf, f * kn), (1, 0), value=True)   # the null key is always visible
        sim = sim.masked_fill(~cm, mask_value)
        attn = sim.softmax(dim=-1, dtype=torch.float32)
        attn = self.dropout(einsum('g h, b h f i j -> b g f i j', self.talking_heads.weight.reshape(h, h), attn))
        out = einsum('b h f i j, b h i j d -> b h f i d', attn, vf)
        out = out.permute(0, 2, 3, 1, 4).reshape(b, -1, h * out.shape[-1])
        out = torch.cat((out_bos, out), dim=1)
        return self.to_out(out[:, :n])


# ---------------------------------------------------------------------------------------------------
# transformer stacks
# ---------------------------------------------------------------------------------------------------

class Transformer(nn.Module):
    """np.py:1071-1182.  forward accepts (and ignores) rotary_pos_emb for the sparse-3DNA decoder;
    for a plain-attention encoder it is routed to the self-attention (superset of the reference,
    whose non-reversible encoder raises TypeError -- quirk Q1)."""

    def __init__(self, *, dim, depth, causal=False, heads=8, dim_head=64, ff_mult=4, cross_attend=False,
                 attn_dropout=0., ff_dropout=0., ff_chunk_size=None, cross_2dna_attn=False, cross_2dna_image_size=None,
                 cross_2dna_kernel_size=3, cross_2dna_dilations=(1,), sparse_3dna_attn=False, sparse_3dna_kernel_size=3,
                 sparse_3dna_video_shape=None, sparse_3dna_query_num_frames_chunk=None, sparse_3dna_dilations=(1,),
                 sparse_3dna_rel_pos_bias=False, shift_video_tokens=False, rotary_pos_emb=False):
        super().__init__()
        assert not (sparse_3dna_attn and not exists(sparse_3dna_video_shape)), 'sparse_3dna_video_shape must be defined if turned on'
        assert not (cross_2dna_attn and not exists(cross_2dna_image_size)), 'cross_2dna_image_size must be defined'
        self.layers = MList([])
        for ind in range(depth):
            if sparse_3dna_attn:
                dilation = sparse_3dna_dilations[ind % len(sparse_3dna_dilations)]
                self_attn = Sparse3DNA(dim=dim, heads=heads, dim_head=dim_head, causal=causal,
                                       kernel_size=sparse_3dna_kernel_size, dilation=dilation,
                                       video_shape=sparse_3dna_video_shape,
                                       query_num_frames_chunk=sparse_3dna_query_num_frames_chunk,
                                       rel_pos_bias=sparse_3dna_rel_pos_bias)
            else:
                self_attn = Attention(dim=dim, heads=heads, dim_head=dim_head, causal=causal, dropout=attn_dropout)
            cross_attn = None
            if cross_attend:
                if cross_2dna_attn:
                    cross_attn = SparseCross2DNA(dim=dim, heads=heads, dim_head=dim_head, dropout=attn_dropout,
                                                 image_size=cross_2dna_image_size, kernel_size=cross_2dna_kernel_size,
                                                 dilation=cross_2dna_dilations[ind % len(cross_2dna_dilations)])
                else:
                    cross_attn = Attention(dim=dim, heads=heads, dim_head=dim_head, dropout=attn_dropout)
            ff = FeedForward(dim=dim, mult=ff_mult, dropout=ff_dropout, chunk_size=ff_chunk_size)
            if sparse_3dna_attn and shift_video_tokens:
                fmap_size = sparse_3dna_video_shape[-1]
                self_attn = ShiftVideoTokens(self_attn, image_size=fmap_size)
                ff = ShiftVideoTokens(ff, image_size=fmap_size)
            self.layers.append(MList([
                SandwichNorm(dim=dim, fn=self_attn),
                SandwichNorm(dim=dim, fn=cross_attn) if cross_attend else None,
                SandwichNorm(dim=dim, fn=ff)
            ]))
        self.norm = StableLayerNorm(dim)

    chain_blocks = True          # fuse each block's post-norm with the next block's pre-norm (A/B switch for tests / probes)

    def forward_layers(self, x, mask=None, context=None, context_mask=None, rotary_pos_emb=None):
        # the sub-blocks in execution order: (block, kwargs of the fused node, kwargs of the plain call, fusable (fn, fmap) or None)
        n, cuda = x.shape[1], x.is_cuda
        calls = []
        for attn, cross_attn, ff in self.layers:
            calls.append((attn, dict(mask=mask, rotary_pos_emb=rotary_pos_emb), dict(mask=mask, rotary_pos_emb=rotary_pos_emb),
                          attn._inner(seq_len=n) if cuda else None))
            if exists(cross_attn):
                calls.append((cross_attn, dict(context=context, context_mask=context_mask),
                              dict(context=context, mask=mask, context_mask=context_mask),
                              cross_attn._inner(context) if cuda else None))
            calls.append((ff, {}, {}, ff._inner() if cuda else None))
        if cuda and K.mixed():
            # fp16 range verdicts of every weight the 'bf16x3-fwd' forward may run in fp16, in ONE device -> host transfer per step
            ws = []
            for _, _, _, inner in calls:
                fn = inner[0] if inner is not None else None
                if isinstance(fn, FeedForward):
                    ws += [fn.net[0].weight, fn.net[3].weight]
                elif isinstance(fn, Sparse3DNA):
                    ws += [fn.to_q.weight, fn.to_kv.weight, fn.to_out.weight]
                elif isinstance(fn, Attention):
                    ws += [fn.to_q.weight, fn.to_out.weight]
            ops.f16_ranges_prefetch(ws)
        handoff = None
        for i, (block, fused_kw, plain_kw, inner) in enumerate(calls):
            if inner is None:
                x, handoff = block(x, **plain_kw) + x, None
                continue
            # chain consecutive fused blocks: this block's post-norm kernel also writes the next block's pre-norm output
            nxt = calls[i + 1] if self.chain_blocks and i + 1 < len(calls) and calls[i + 1][3] is not None else None
            out = {}
            x = block.fused_residual(x, chain=(handoff, nxt[0] if nxt else None, nxt[3][1] if nxt else None, out, nxt[1] if nxt else {}),
                                     **fused_kw)
            handoff = out if out else None
        return x

    def forward(self, x, mask=None, context=None, context_mask=None, rotary_pos_emb=None):
        x = self.forward_layers(x, mask=mask, context=context, context_mask=context_mask, rotary_pos_emb=rotary_pos_emb)
        return self.norm(x)


# reversible plumbing (rev.py) ------------------------------------------------------------------------

def route_args(router, args, depth):
    """per block d the keyword sets of its two halves: `router[name][d] = (to_f, to_g)` says whether f / g of block d receives the
    keyword `name` (the routing of rev.py:8-17; keywords without a route are dropped)"""
    routed = []
    for d in range(depth):
        routed.append(tuple({k: v for k, v in args.items() if k in router and router[k][d][half]} for half in (0, 1)))
    return routed


class Deterministic(nn.Module):
    """rev.py:20-50: keeps the `.net` level of the state-dict key hierarchy.  No dropout on this path,
    so there is no RNG to record/replay."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, *args, **kwargs):
        return self.net(*args, **kwargs)


class ReversibleBlock(nn.Module):
    """rev.py:54-106: y1 = x1 + f(x2); y2 = x2 + g(y1)."""

    def __init__(self, f, g):
        super().__init__()
        self.f = Deterministic(f)
        self.g = Deterministic(g)

    def forward(self, x1, x2, f_args={}, g_args={}):
        f, g = self.f.net, self.g.net
        if isinstance(f, SandwichNorm) and x1.is_cuda and f._inner(f_args.get('context'), seq_len=x2.shape[1]) is not None:
            y1 = f.fused_residual(x2, resid=x1, **{k: f_args.get(k) for k in ('context', 'context_mask', 'mask', 'rotary_pos_emb')})
        else:
            y1 = x1 + f(x2, **f_args)
        if isinstance(g, SandwichNorm) and g._inner() is not None and x1.is_cuda:
            y2 = g.fused_residual(y1, resid=x2)
        else:
            y2 = x2 + g(y1, **g_args)
        return y1, y2

    def backward_pass(self, y1, y2, dy1, dy2, f_args={}, g_args={}):
        """inputs and input-gradients of this block from its outputs and output-gradients (the role of rev.py:77-106);
        parameter gradients of f and g are accumulated by the two inner autograd calls"""
        f, g = self.f.net, self.g.net
        fuse_f = isinstance(f, SandwichNorm) and y1.is_cuda and f._inner(f_args.get('context'), seq_len=y1.shape[1]) is not None
        fuse_g = isinstance(g, SandwichNorm) and g._inner() is not None and y1.is_cuda
        # g(y1) again.  Fused form: one node whose value is y2 - g(y1) = x2 (the post-norm kernel subtracts: no negation passes over the
        # stream -- four of them per block, 6 % of the cfg-4 step, in the (-y2) + g(y1) = -x2 form of rounds 2-4; the same bits) and whose
        # gradient w.r.t. y1 and g's parameters is g's.
        with torch.enable_grad():
            y1g = y1.detach().requires_grad_(True)
            if fuse_g:
                x2n = g.fused_residual(y1g, resid=y2, minus=True)
                x2n.grad_fn.dx_add = dy1                  # (the node's backward returns dy1 + dL/dy1: ops.SandwichBlockFn.backward)
                torch.autograd.backward(x2n, dy2)
            else:
                gy1 = g(y1g, **g_args)
                torch.autograd.backward(gy1, dy2)
        with torch.no_grad():
            x2 = x2n.detach() if fuse_g else y2 - gy1.detach()
            dx1 = y1g.grad if fuse_g else dy1 + y1g.grad
        with torch.enable_grad():
            x2g = x2.detach().requires_grad_(True)
            if fuse_f:
                x1n = f.fused_residual(x2g, resid=y1, minus=True, **{k: f_args.get(k) for k in ('context', 'context_mask', 'mask', 'rotary_pos_emb')})
                x1n.grad_fn.dx_add = dy2
                torch.autograd.backward(x1n, dx1)
            else:
                fx2 = f(x2g, **f_args)
                torch.autograd.backward(fx2, dx1)
        with torch.no_grad():
            x1 = x1n.detach() if fuse_f else y1 - fx2.detach()
            dx2 = x2g.grad if fuse_f else dy2 + x2g.grad
        return x1, x2, dx1, dx2


class _ReversibleStackFn(torch.autograd.Function):
    """O(1)-activation-memory execution of a stack of reversible blocks (the role of rev.py:108-124): the forward keeps only
    the output pair; the backward walks the blocks in reverse, reconstructing each block's inputs from its outputs
    (x2 = y2 - g(y1), x1 = y1 - f(x2)) and back-propagating through one freshly recomputed f / g at a time.  Parameter (and
    context) gradients accumulate through the inner autograd calls, as in the reference."""

    @staticmethod
    def forward(ctx, x, context, seq, args):
        # `context` is an explicit input so that its gradient leaves this node ONCE (summed over the cross-attention blocks)
        # instead of re-entering the text encoder's graph from every block, as the reference's retain_graph=True does
        x1 = x2 = x.detach()
        cdet = context.detach() if context is not None else None
        args = [tuple({k: (cdet if k == 'context' else v) for k, v in a.items()} for a in pair) for pair in args]
        for block, (f_args, g_args) in zip(seq.blocks, args):
            x1, x2 = block(x1, x2, f_args=f_args, g_args=g_args)
        ctx.seq, ctx.args, ctx.cdet = seq, args, cdet
        ctx.save_for_backward(x1, x2)
        return x1 + x2

    @staticmethod
    def backward(ctx, dy):
        y1, y2 = ctx.saved_tensors
        dy1 = dy2 = dy
        dctx = None
        for block, (f_args, g_args) in reversed(list(zip(ctx.seq.blocks, ctx.args))):
            leaf = None
            if ctx.cdet is not None and ('context' in f_args or 'context' in g_args):
                leaf = ctx.cdet.detach().requires_grad_(True)
                f_args = {k: (leaf if k == 'context' else v) for k, v in f_args.items()}
                g_args = {k: (leaf if k == 'context' else v) for k, v in g_args.items()}
            y1, y2, dy1, dy2 = block.backward_pass(y1, y2, dy1, dy2, f_args=f_args, g_args=g_args)
            if leaf is not None and leaf.grad is not None:
                dctx = leaf.grad if dctx is None else dctx + leaf.grad
        return dy1 + dy2, dctx, None, None


class ReversibleSequence(nn.Module):
    """rev.py:126-142.  Same arithmetic as the reference (x -> (x, x); blocks; sum of the halves).  In training the stack runs
    through _ReversibleStackFn (activations of one block at a time); `memory_efficient = False` keeps every activation in the
    autograd graph instead (bit-for-bit the same forward, no recomputation error in the backward)."""

    memory_efficient = True

    def __init__(self, blocks, args_route={}):
        super().__init__()
        self.args_route = args_route
        self.blocks = nn.ModuleList([ReversibleBlock(f=f, g=g) for f, g in blocks])

    def forward(self, x, **kwargs):
        args = route_args(self.args_route, kwargs, len(self.blocks))
        if self.memory_efficient and torch.is_grad_enabled() and x.is_cuda and \
                (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _ReversibleStackFn.apply(x, kwargs.get('context'), self, args)
        x1, x2 = x, x
        for block, (f_args, g_args) in zip(self.blocks, args):
            x1, x2 = block(x1, x2, f_args=f_args, g_args=g_args)
        return x1 + x2


class ReversibleTransformer(nn.Module):
    """np.py:1184-1295 (every parameter appears twice in the state dict: layers.* and net.blocks.*)."""

    def __init__(self, *, dim, depth, causal=False, heads=8, dim_head=64, ff_mult=4, cross_attend=False,
                 attn_dropout=0., ff_dropout=0., ff_chunk_size=None, cross_2dna_attn=False, cross_2dna_image_size=None,
                 cross_2dna_kernel_size=3, cross_2dna_dilations=(1,), sparse_3dna_attn=False, sparse_3dna_kernel_size=3,
                 sparse_3dna_video_shape=None, sparse_3dna_query_num_frames_chunk=None, sparse_3dna_dilations=(1,),
                 sparse_3dna_rel_pos_bias=False, shift_video_tokens=False, rotary_pos_emb=False):
        super().__init__()
        assert not (sparse_3dna_attn and not exists(sparse_3dna_video_shape)), 'sparse_3dna_video_shape must be defined if turned on'
        self.layers = MList([])
        for ind in range(depth):
            if sparse_3dna_attn:
                dilation = sparse_3dna_dilations[ind % len(sparse_3dna_dilations)]
                image_size = sparse_3dna_video_shape[-1]
                self_attn = Sparse3DNA(dim=dim, heads=heads, dim_head=dim_head, causal=causal,
                                       kernel_size=sparse_3dna_kernel_size, dilation=dilation,
                                       video_shape=sparse_3dna_video_shape,
                                       query_num_frames_chunk=sparse_3dna_query_num_frames_chunk,
                                       rel_pos_bias=sparse_3dna_rel_pos_bias)
            else:
                image_size = None
                self_attn = Attention(dim=dim, heads=heads, dim_head=dim_head, causal=causal, dropout=attn_dropout)
            wrapper_fn = partial(ShiftVideoTokens, image_size=image_size, shift_space=sparse_3dna_attn and shift_video_tokens)
            self.layers.append(MList([
                SandwichNorm(dim=dim, fn=wrapper_fn(self_attn)),
                SandwichNorm(dim=dim, fn=wrapper_fn(FeedForward(dim=dim, mult=ff_mult, dropout=ff_dropout, chunk_size=ff_chunk_size)))
            ]))
            if not cross_attend:
                continue
            if cross_2dna_attn:
                cross_attn = SparseCross2DNA(dim=dim, heads=heads, dim_head=dim_head, dropout=attn_dropout,
                                             image_size=cross_2dna_image_size, kernel_size=cross_2dna_kernel_size,
                                             dilation=cross_2dna_dilations[ind % len(cross_2dna_dilations)])
            else:
                cross_attn = Attention(dim=dim, heads=heads, dim_head=dim_head, dropout=attn_dropout)
            self.layers.append(MList([
                SandwichNorm(dim=dim, fn=cross_attn),
                SandwichNorm(dim=dim, fn=wrapper_fn(FeedForward(dim=dim, mult=ff_mult, dropout=ff_dropout, chunk_size=ff_chunk_size)))
            ]))
        attn_context_layer = ((True, False),) if cross_attend else tuple()
        route_attn = ((True, False), *attn_context_layer) * depth
        route_context = ((False, False), *attn_context_layer) * depth
        context_route_map = {'context': route_context, 'context_mask': route_context} if cross_attend else {}
        attn_route_map = {'mask': route_attn, 'rotary_pos_emb': route_attn}
        self.net = ReversibleSequence(self.layers, args_route={**context_route_map, **attn_route_map})
        self.norm = StableLayerNorm(dim)
        for m in self.modules():                 # a reversible block's forward runs twice per step without RNG replay: no kept dropout mask here
            if isinstance(m, FeedForward):
                m._no_hip_dropout = True

    def forward_layers(self, x, **kwargs):
        return self.net(x, **kwargs)

    def forward(self, x, **kwargs):
        return self.norm(self.net(x, **kwargs))


# ---------------------------------------------------------------------------------------------------
# embeddings
# ---------------------------------------------------------------------------------------------------

def frac_gradient(t, frac):
    return t * frac + t.detach() * (1 - frac)


class Embedding(nn.Module):
    """np.py:1659-1671"""

    def __init__(self, *shape, frac_gradient=1.):
        super().__init__()
        self.frac_gradient = frac_gradient
        self.embed = nn.Embedding(*shape)

    def forward(self, x):
        x = self.embed(x)
        if self.training and self.frac_gradient < 1:
            x = frac_gradient(x, self.frac_gradient)
        return x


class AxialPositionalEmbedding(nn.Module):
    """np.py:1675-1709"""

    def __init__(self, dim, *, shape):
        super().__init__()
        shape = tuple(filter(lambda t: t > 1, shape))
        self.dim = dim
        self.shape = shape
        self.num_axials = len(shape)
        for axial_ind, axial_len in enumerate(shape):
            setattr(self, f'axial{axial_ind + 1}', nn.Parameter(torch.randn(axial_len, dim)))

    def forward(self, *, flatten=True):
        positions = None
        for axial_ind in range(self.num_axials):
            axial_pos = getattr(self, f'axial{axial_ind + 1}')
            if not exists(positions):
                positions = axial_pos
                continue
            positions = positions.unsqueeze(-2) + axial_pos
        if flatten:
            positions = positions.reshape(-1, positions.shape[-1])
        return positions


# ---------------------------------------------------------------------------------------------------
# main class
# ---------------------------------------------------------------------------------------------------

class NUWA(nn.Module):
    """np.py:1723-1964: identical constructor kwargs, forward() and generate() signatures."""

    def __init__(self, *, dim, vae=None, image_size=None, max_video_frames=5, text_num_tokens=49408,
                 text_max_seq_len=256, text_enc_depth=6, text_enc_dim_head=64, text_enc_heads=8, text_rotary_pos_emb=True,
                 enc_reversible=False, dec_depth=6, dec_dim_head=64, dec_heads=8, dec_reversible=False, attn_dropout=0.,
                 ff_dropout=0., ff_chunk_size=None, embed_gradient_frac=0.2, shift_video_tokens=True,
                 sparse_3dna_kernel_size=3, sparse_3dna_query_num_frames_chunk=None, sparse_3dna_dilation=1,
                 sparse_3dna_rel_pos_bias=False):
        super().__init__()
        assert exists(vae) ^ exists(image_size), 'either VAE or image size must be specified'
        assert exists(vae), 'a VAE must be passed (the reference dereferences vae.num_layers unconditionally, quirk Q3)'
        self.vae = vae.copy_for_eval()
        image_size = vae.image_size
        vae_num_layers = vae.num_layers
        num_image_tokens = vae.codebook_size
        self.text_max_seq_len = text_max_seq_len
        self.text_embedding = Embedding(text_num_tokens, dim, frac_gradient=embed_gradient_frac)
        self.text_abs_pos_emb = Embedding(text_max_seq_len, dim) if not text_rotary_pos_emb else None
        self.text_rotary_pos_emb = RotaryEmbedding(dim=min(32, text_enc_dim_head)) if text_rotary_pos_emb else None
        enc_transformer_klass = Transformer if not enc_reversible else ReversibleTransformer
        self.text_transformer = enc_transformer_klass(dim=dim, depth=text_enc_depth, heads=text_enc_heads,
                                                      dim_head=text_enc_dim_head, attn_dropout=attn_dropout,
                                                      ff_dropout=ff_dropout, rotary_pos_emb=text_rotary_pos_emb)
        self.video_bos = nn.Parameter(torch.randn(dim))
        self.image_embedding = Embedding(num_image_tokens, dim, frac_gradient=embed_gradient_frac)
        fmap_size = image_size // (2 ** vae_num_layers)
        self.video_fmap_size = fmap_size
        self.max_video_frames = max_video_frames
        video_shape = (max_video_frames, fmap_size, fmap_size)
        self.video_shape = video_shape
        self.video_pos_emb = AxialPositionalEmbedding(dim, shape=video_shape)
        sparse_3dna_dilations = tuple(range(1, sparse_3dna_dilation + 1)) if not isinstance(sparse_3dna_dilation, (list, tuple)) else sparse_3dna_dilation
        dec_transformer_klass = Transformer if not dec_reversible else ReversibleTransformer
        self.video_transformer = dec_transformer_klass(
            dim=dim, depth=dec_depth, heads=dec_heads, dim_head=dec_dim_head, causal=True, cross_attend=True,
            attn_dropout=attn_dropout, ff_dropout=ff_dropout, ff_chunk_size=ff_chunk_size,
            shift_video_tokens=shift_video_tokens, sparse_3dna_video_shape=video_shape, sparse_3dna_attn=True,
            sparse_3dna_kernel_size=sparse_3dna_kernel_size, sparse_3dna_dilations=sparse_3dna_dilations,
            sparse_3dna_query_num_frames_chunk=sparse_3dna_query_num_frames_chunk,
            sparse_3dna_rel_pos_bias=sparse_3dna_rel_pos_bias)
        self.to_logits = nn.Linear(dim, num_image_tokens, bias=False)
        self._cache = ops.WeightCache()

    generate_use_cache = True        # key/value-cached generate() (decode.py); False = the reference's recompute loop
    generate_use_graph = True        # replay each token's decoder work as one captured HIP graph

    # -- text side (adjacent, row f1) ------------------------------------------------------------
    def embed_text(self, text, mask=None):
        batch, seq_len, device = *text.shape, text.device
        assert seq_len <= self.text_max_seq_len, 'your input text has a greater length than what was designated on initialization'
        tokens = self.text_embedding(text)
        if exists(self.text_abs_pos_emb):
            pos_emb = self.text_abs_pos_emb(torch.arange(seq_len, device=device))
            tokens = tokens + pos_emb[None]
        rotary_pos_emb = None
        if exists(self.text_rotary_pos_emb):
            rotary_pos_emb = self.text_rotary_pos_emb(seq_len, device=device)
        return self.text_transformer(tokens, mask=mask, rotary_pos_emb=rotary_pos_emb)

    # -- decoder side (the hot path) ----------------------------------------------------------------
    def embed_video(self, ids_in):
        """ids_in (b, m) -> (b, m+1, dim): <bos> + positional + token embedding (np.py:1940-1944)"""
        pe = self.video_pos_emb
        if ids_in.shape[1] == 0:                                   # first step of generate(): the sequence is just <bos>
            return self.video_bos[None, None].expand(ids_in.shape[0], 1, -1).contiguous()
        if ids_in.is_cuda and pe.num_axials == 3:
            frac = self.image_embedding.frac_gradient if self.training else 1.
            return ops.EmbedAssembleFn.apply(ids_in, self.image_embedding.embed.weight, pe.axial1, pe.axial2, pe.axial3,
                                             self.video_bos, self.video_shape, float(frac))
        emb = self.image_embedding(ids_in)
        emb = pe()[:ids_in.shape[1]] + emb
        bos = self.video_bos[None, None].expand(ids_in.shape[0], 1, -1)
        return torch.cat((bos, emb), dim=1)

    def decode_hidden(self, frame_embeddings, text_embeds, text_mask):
        """decoder layers WITHOUT the final StableLayerNorm (fused into the logits/loss node)"""
        return self.video_transformer.forward_layers(frame_embeddings, context=text_embeds, context_mask=text_mask)

    def _final(self, hidden, targets=None):
        nrm = self.video_transformer.norm.norm
        if hidden.is_cuda:
            if exists(targets):
                return ops.LogitsLossFn.apply(hidden, targets, nrm.weight, nrm.bias, self.to_logits.weight, self._cache)
            return ops.LogitsFn.apply(hidden, nrm.weight, nrm.bias, self.to_logits.weight, self._cache)
        raise RuntimeError('nuwa_pytorch_amd: the decoder path needs a HIP device; there is no CPU fallback')

    def _guided_last_logits(self, ids, context, context_mask, cond_scale):
        """logits of the NEXT token after `ids` by recomputing the whole prefix -- the reference's decoding algorithm
        (np.py:1883-1903): with guidance, a second pass is fed the first pass's NORMED output and sees no condition"""
        hidden = self.decode_hidden(self.embed_video(ids), context, context_mask)
        logits = self._final(hidden[:, -1:].contiguous())
        if cond_scale != 1:
            blind = self.decode_hidden(self.video_transformer.norm(hidden), context, torch.zeros_like(context_mask))
            base = self._final(blind[:, -1:].contiguous())
            logits = base + (logits - base) * cond_scale
        return logits[:, -1]

    def _ids_to_frames(self, ids, decode_max_batchsize):
        """token ids (b, frames * fmap^2) -> frames (b, frames, c, H, W) through the VAE decoder (np.py:1910-1915)"""
        batch, fs = ids.shape[0], self.video_fmap_size
        codes = self.vae.codes_for_decoder(ids)
        codes = codes.reshape(batch, -1, fs, fs, codes.shape[-1]).permute(0, 1, 4, 2, 3).reshape(-1, codes.shape[-1], fs, fs)
        decode = self.vae._hip_decode if codes.is_cuda else self.vae.decode
        frames = map_in_chunks(codes.contiguous(), decode, chunks=decode_max_batchsize)
        return frames.reshape(batch, -1, *frames.shape[1:])

    @torch.no_grad()
    @eval_decorator
    def generate(self, *, text, filter_thres=0.9, temperature=1., decode_max_batchsize=10, cond_scale=2., num_frames=None):
        """np.py:1841-1915.  The reference recomputes the whole prefix (twice) per token; here each token costs one new decoder
        row against per-layer key/value caches (decode.GuidedStepper, row f3) whenever the sequence fits the video shape and every
        decoder block (Transformer or ReversibleTransformer) is on the single-row kernels -- otherwise the reference's recompute algorithm runs on the same kernels."""
        batch, device = text.shape[0], text.device
        text_mask = text != 0
        text_embeds = self.embed_text(text, mask=text_mask)
        tpf = self.video_fmap_size ** 2
        total = tpf * default(num_frames, self.max_video_frames)
        ids = torch.empty((batch, 0), device=device, dtype=torch.long)
        cached = self.generate_use_cache and text.is_cuda and total <= tpf * self.max_video_frames
        if cached:
            from .decode import GuidedStepper
            try:                                 # plain and reversible decoder alike; a block outside the single-row kernels -> recompute
                stepper = GuidedStepper(self, text_embeds, text_mask, total, cond_scale, graph=self.generate_use_graph)
            except NotImplementedError:
                cached = False
        if cached:
            pos_table = self.video_pos_emb()
            row = self.video_bos[None].expand(batch, -1)
        for t in range(total):
            if cached:
                logits = stepper(row)
            else:
                logits = self._guided_last_logits(lookback_window(ids, tpf, self.max_video_frames), text_embeds, text_mask, cond_scale)
            token = sample_top_fraction(logits, filter_thres, temperature)
            ids = torch.cat((ids, token[:, None]), dim=1)
            if cached:
                row = self.image_embedding(token) + pos_table[t]
        self.last_generated_ids = ids                      # (b, frames * fmap^2) token ids behind the returned frames
        return self._ids_to_frames(ids, decode_max_batchsize)

    def forward(self, *, text, video=None, return_loss=False, cond_dropout_prob=0.2):
        batch, seq_len, frames, device = *text.shape, video.shape[1], text.device
        text_mask = text != 0
        text_embeds = self.embed_text(text, mask=text_mask)
        if video.dtype == torch.long:
            frame_indices = video
        else:
            assert frames == self.max_video_frames, f'you must give the full video frames ({self.max_video_frames}) during training'
            assert exists(self.vae), 'VAE must be passed in if you wish for video to be encoded to ids automatically'
            frame_indices = self.vae.get_video_indices(video)
        frame_indices = frame_indices.reshape(batch, -1)
        frame_indices_input = frame_indices[:, :-1] if return_loss else frame_indices
        frame_embeddings = self.embed_video(frame_indices_input)
        if self.training and cond_dropout_prob > 0:
            text_mask = text_mask & ~bernoulli_rows(batch, cond_dropout_prob, device)[:, None]
        if frame_embeddings.is_cuda and K.mixed() and ((return_loss and ops.FUSE_LINEAR_CE_X3) or K.proj_f16x2()):
            ops.f16_prefetch_also(self.to_logits.weight)       # judged with the stack's weights: one device -> host transfer per step
        hidden = self.decode_hidden(frame_embeddings, text_embeds, text_mask)
        if not return_loss:
            return self._final(hidden)
        return self._final(hidden, frame_indices)


class NUWASketch(nn.Module):
    """np.py:2297-2571 (row f4): identical constructor kwargs, forward() and generate() signatures.  The video decoder is the
    same 3DNA stack as NUWA's (libamdnuwa kernels for Sparse3DNA, FeedForward, the norms, logits + loss); its cross-attention
    is SparseCross2DNA over the sketch tokens (amdnuwa_cross2dna_* in training, the single-query kernels in the cached generate())
    and the sketch encoder is a plain (or, optionally, non-causal 3DNA) Transformer."""

    def __init__(self, *, vae, sketch_vae, dim, image_size, max_video_frames=5, sketch_max_video_frames=2, sketch_enc_depth=6,
                 sketch_enc_dim_head=64, sketch_enc_heads=8, sketch_enc_use_sparse_3dna=False, enc_reversible=False, dec_depth=6,
                 dec_dim_head=64, dec_heads=8, dec_reversible=False, attn_dropout=0., ff_dropout=0., ff_chunk_size=None,
                 embed_gradient_frac=0.2, shift_video_tokens=True, cross_2dna_kernel_size=3, cross_2dna_dilation=1,
                 sparse_3dna_kernel_size=3, sparse_3dna_dilation=1, sparse_3dna_query_num_frames_chunk=None):
        super().__init__()
        self.image_size = image_size
        self.sketch_vae = sketch_vae
        sketch_fmap_size = image_size // (2 ** sketch_vae.num_layers)
        sketch_shape = (sketch_max_video_frames, sketch_fmap_size, sketch_fmap_size)
        self.sketch_max_video_frames = sketch_max_video_frames
        self.sketch_embedding = Embedding(sketch_vae.codebook_size, dim, frac_gradient=embed_gradient_frac)
        self.sketch_pos_emb = AxialPositionalEmbedding(dim, shape=sketch_shape)
        sparse_3dna_dilations = tuple(range(1, sparse_3dna_dilation + 1)) if not isinstance(sparse_3dna_dilation, (list, tuple)) else sparse_3dna_dilation
        enc_transformer_klass = Transformer if not enc_reversible else ReversibleTransformer
        self.sketch_transformer = enc_transformer_klass(
            dim=dim, depth=sketch_enc_depth, heads=sketch_enc_heads, dim_head=sketch_enc_dim_head, attn_dropout=attn_dropout,
            ff_dropout=ff_dropout, shift_video_tokens=shift_video_tokens, sparse_3dna_video_shape=sketch_shape,
            sparse_3dna_kernel_size=sparse_3dna_kernel_size, sparse_3dna_dilations=sparse_3dna_dilations,
            sparse_3dna_query_num_frames_chunk=sparse_3dna_query_num_frames_chunk, sparse_3dna_attn=sketch_enc_use_sparse_3dna)
        self.vae = vae.copy_for_eval()
        self.video_bos = nn.Parameter(torch.randn(dim))
        self.image_embedding = Embedding(vae.codebook_size, dim, frac_gradient=embed_gradient_frac)
        fmap_size = image_size // (2 ** vae.num_layers)
        assert fmap_size == sketch_fmap_size, 'feature map size of video must be equal to the feature map size of sketches (VAEs must have same number of layers)'
        self.video_fmap_size = fmap_size
        self.max_video_frames = max_video_frames
        self.video_shape = (max_video_frames, fmap_size, fmap_size)
        self.video_pos_emb = AxialPositionalEmbedding(dim, shape=self.video_shape)
        cross_2dna_dilations = tuple(range(1, cross_2dna_dilation + 1)) if not isinstance(cross_2dna_dilation, (list, tuple)) else cross_2dna_dilation
        dec_transformer_klass = Transformer if not dec_reversible else ReversibleTransformer
        self.video_transformer = dec_transformer_klass(
            dim=dim, depth=dec_depth, heads=dec_heads, dim_head=dec_dim_head, causal=True, cross_attend=True, cross_2dna_attn=True,
            cross_2dna_image_size=fmap_size, cross_2dna_kernel_size=cross_2dna_kernel_size, cross_2dna_dilations=cross_2dna_dilations,
            attn_dropout=attn_dropout, ff_dropout=ff_dropout, ff_chunk_size=ff_chunk_size, shift_video_tokens=shift_video_tokens,
            sparse_3dna_video_shape=self.video_shape, sparse_3dna_kernel_size=sparse_3dna_kernel_size,
            sparse_3dna_dilations=sparse_3dna_dilations, sparse_3dna_query_num_frames_chunk=sparse_3dna_query_num_frames_chunk,
            sparse_3dna_attn=True)
        self.to_logits = nn.Linear(dim, vae.codebook_size, bias=False)
        self._cache = ops.WeightCache()

    generate_use_cache = True               # key/value-cached generate() (decode.py); False = the reference's recompute loop
    generate_use_graph = True               # replay each token's decoder work (rows >= 1) as one captured HIP graph
    embed_video = NUWA.embed_video          # <bos> + positional + token embedding, one libamdnuwa node
    _final = NUWA._final                    # final StableLayerNorm + logits (+ cross entropy), fused
    _guided_last_logits = NUWA._guided_last_logits
    _ids_to_frames = NUWA._ids_to_frames

    def decode_hidden(self, frame_embeddings, sketch_embeds, context_mask):
        return self.video_transformer.forward_layers(frame_embeddings, context=sketch_embeds, context_mask=context_mask)

    def embed_sketch(self, sketch, mask=None):
        batch, frames, device = sketch.shape[0], sketch.shape[1], sketch.device
        if exists(mask):
            assert mask.shape[:2] == (batch, frames), 'sketch mask must be in shape of (batch x frame)'
        sketch_indices = self.sketch_vae.get_video_indices(sketch).reshape(batch, -1)
        sketch_tokens = self.sketch_embedding(sketch_indices)
        num_tokens = sketch_tokens.shape[1]
        sketch_tokens = sketch_tokens + self.sketch_pos_emb()[:num_tokens]
        if exists(mask):
            mask = mask[:, :, None].expand(-1, -1, num_tokens // frames).reshape(batch, num_tokens)
        else:
            mask = torch.ones((batch, num_tokens), dtype=torch.bool, device=device)
        return self.sketch_transformer(sketch_tokens, mask=mask), mask

    @torch.no_grad()
    @eval_decorator
    def generate(self, *, sketch, sketch_mask=None, filter_thres=0.9, temperature=1., decode_max_batchsize=10, cond_scale=2.,
                 num_frames=None):
        """np.py:2438-2511.  The reference recomputes the whole prefix per token (and, for guidance, feeds the normed conditioned
        output to a sketch-masked second pass).  Every decoder stage is row-causal or row-wise -- SparseCross2DNA looks at the sketch
        only -- so each token costs one new decoder row against per-layer caches (decode.GuidedStepper, as NUWA.generate) whenever the
        sequence fits the video shape and every block has a single-row path; otherwise the recompute algorithm runs on the same kernels."""
        if sketch.ndim == 4:
            sketch = sketch[:, None]
        batch, device = sketch.shape[0], sketch.device
        sketch_embeds, context_mask = self.embed_sketch(sketch, mask=sketch_mask)
        tpf = self.video_fmap_size ** 2
        total = tpf * default(num_frames, self.max_video_frames)
        ids = torch.empty((batch, 0), device=device, dtype=torch.long)
        cached = self.generate_use_cache and sketch.is_cuda and total <= tpf * self.max_video_frames
        if cached:
            from .decode import GuidedStepper
            try:
                stepper = GuidedStepper(self, sketch_embeds, context_mask, total, cond_scale, graph=self.generate_use_graph)
            except NotImplementedError:
                cached = False
        if cached:
            pos_table = self.video_pos_emb()
            row = self.video_bos[None].expand(batch, -1)
        for t in range(total):
            if cached:
                logits = stepper(row)
            else:
                logits = self._guided_last_logits(lookback_window(ids, tpf, self.max_video_frames), sketch_embeds, context_mask, cond_scale)
            token = sample_top_fraction(logits, filter_thres, temperature)
            ids = torch.cat((ids, token[:, None]), dim=1)
            if cached:
                row = self.image_embedding(token) + pos_table[t]
        self.last_generated_ids = ids
        return self._ids_to_frames(ids, decode_max_batchsize)

    def forward(self, *, sketch, sketch_mask=None, video=None, return_loss=False, cond_dropout_prob=0.2):
        if sketch.ndim == 4:                              # one sketch frame
            sketch = sketch[:, None]
        batch, sketch_frames, device = sketch.shape[0], sketch.shape[1], sketch.device
        assert sketch.shape[-1] == self.image_size, 'sketch image size must be equal'
        assert sketch_frames <= self.sketch_max_video_frames, 'sketch frames must be less than max sketch video frames'
        sketch_embeds, context_mask = self.embed_sketch(sketch, mask=sketch_mask)
        if video.dtype == torch.long:                     # pre-tokenised video (as NUWA.forward accepts; the reference only takes frames)
            frame_indices = video
        else:
            assert video.shape[1] == self.max_video_frames, f'you must give the full video frames ({self.max_video_frames}) during training'
            frame_indices = self.vae.get_video_indices(video)
        frame_indices = frame_indices.reshape(batch, -1)
        frame_embeddings = self.embed_video(frame_indices[:, :-1] if return_loss else frame_indices)
        if self.training and cond_dropout_prob > 0:
            # the reference multiplies `sketch_mask` in place AFTER the decoder mask was derived from it (np.py:2551-2555), which
            # drops nothing (and raises when sketch_mask is None); the evident intent -- drop the condition -- is applied here
            context_mask = context_mask & ~bernoulli_rows(batch, cond_dropout_prob, device)[:, None]
        hidden = self.decode_hidden(frame_embeddings, sketch_embeds, context_mask)
        if not return_loss:
            return self._final(hidden)
        return self._final(hidden, frame_indices)


# BASELINE cfg 5 (video + audio dual decoder) lives in its own module; re-exported here so that the names resolve where the
# reference defines them
from .video_audio import (ShiftAudioTokens, SparseCausal2DNA, CrossModalityCrossAttention, DualModalityDecoder,  # noqa: E402
                          ReversibleDualModalityDecoder, NUWAVideoAudio)
