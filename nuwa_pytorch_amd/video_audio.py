"""BASELINE cfg 5: the video + audio dual decoder (reference nuwa_pytorch.py = np.py: ShiftAudioTokens 157-183,
SparseCausal2DNA 615-759, CrossModalityCrossAttention 908-1067, DualModalityDecoder 1299-1487, NUWAVideoAudio 1968-2293).

Same class names, constructor kwargs, forward() signatures and state-dict keys as the reference.  What runs where:
  * video tower (Sparse3DNA incl. rel_pos_bias, text cross-attention, GEGLU FF, token shift), the text cross-attention and
    the FeedForwards of the audio tower, every LayerNorm, and both StableLayerNorm + logits + cross-entropy heads go through
    libamdnuwa (the same fused autograd nodes as NUWA);
  * the audio-only attention pieces map onto the same kernels: the 1-D causal window attention over audio tokens IS a
    Sparse3DNA over a (time, 1, 1) grid with a (k, 1, 1) kernel and the per-tap bias; the one-frame-lagged chunked
    video<->audio attention is the cross-attention kernel with every (sample, frame) pair as one sample, plus a rank-one
    correction for the talking-heads Conv3d bias.  Head sizes the kernels do not cover (and `use_hip = False`) run the torch-op
    formulations kept next to them (index-table gathers, nothing is unfolded); the audio channel shift is a torch op.
`dec_reversible=True` (ReversibleDualModalityDecoder, np.py:1489-1655 + reversible_video_audio.py) runs with the reference's
arithmetic, with the recomputing (O(1)-activation-memory) backward of reversible_video_audio.py.  generate() decodes one new row per
sampled token against per-layer caches (decode.py: both decoders), falling back to the reference's recompute loop only for
configurations outside the single-row kernels.
"""
import torch
import torch.nn.functional as F
from torch import nn, einsum

from . import ops
from .nuwa_pytorch import (MList, exists, default, eval_decorator, bernoulli_rows, SandwichNorm, ShiftVideoTokens, FeedForward, Deterministic,
                           Attention, Sparse3DNA, StableLayerNorm, Transformer, ReversibleTransformer, Embedding,
                           AxialPositionalEmbedding, RotaryEmbedding)

NEG = -torch.finfo(torch.float32).max


class ShiftAudioTokens(nn.Module):
    """np.py:157-183: the first half of the channels of token i is replaced by that of token i-1 (zeros for i = 0); the
    <bos> row takes part like any other token.  (The reference pads to a whole timestep first and crops again: no effect.)"""

    def __init__(self, fn, audio_tokens_per_timestep=1):
        super().__init__()
        self.fn = fn
        self.audio_tokens_per_timestep = audio_tokens_per_timestep

    def forward(self, x, **kwargs):
        half = (x.shape[-1] + 1) // 2                       # chunk(2): the first chunk is the ceil half
        prev = F.pad(x[:, :-1, :half], (0, 0, 1, 0))
        return self.fn(torch.cat((prev, x[..., half:]), dim=-1), **kwargs)


class SparseCausal2DNA(nn.Module):
    """np.py:615-759 with height = 1 (the only way the decoders build it): audio token t (after <bos>) attends <bos> and the
    taps t - (k-1-a)*dilation, a = 0..k-1, that are >= 0; + per-tap, per-head bias (always present: quirk Q20); fp32 softmax;
    talking heads; the <bos> row outputs its own value."""

    def __init__(self, *, dim, height=1, heads=8, dim_head=64, dropout=0., kernel_size=5, dilation=1, rel_pos_bias=False):
        super().__init__()
        if height != 1:
            raise NotImplementedError('SparseCausal2DNA: only height = 1 (one audio token per timestep row) is built')
        inner = heads * dim_head
        self.heads, self.dim_head, self.scale, self.height = heads, dim_head, dim_head ** -0.5, height
        self._cache = ops.WeightCache()
        self.use_hip = True            # False: the torch-op formulation below (kept for head sizes the kernels do not cover)
        self.talking_heads = nn.Conv3d(heads, heads, 1, bias=False)
        self.dropout = nn.Dropout(dropout)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)
        self.kernel_size = (kernel_size, height)
        self.dilation = (dilation, 1)
        self.rel_pos_bias = AxialPositionalEmbedding(heads, shape=self.kernel_size)      # exists(False) is True in the reference

    def _taps(self, m, device):
        """[m, k] index of the key token (0-based among the non-<bos> tokens) of every tap, -1 where it falls before the start"""
        k, dil = self.kernel_size[0], self.dilation[0]
        t = torch.arange(m, device=device)[:, None]
        a = torch.arange(k, device=device)[None, :]
        idx = t - (k - 1 - a) * dil
        return torch.where(idx >= 0, idx, torch.full_like(idx, -1))

    # the same attention IS a Sparse3DNA over a (time, 1, 1) grid with kernel (k, 1, 1) and the per-tap bias: it runs on the
    # libamdnuwa 3DNA kernels (to_qkv split into the q / kv halves they expect; to_out has no bias -> zeros), stand-alone through
    # ops.InnerFn or, inside a SandwichNorm, as the inner stage of the fused block node
    def _hip_ok(self):
        return self.use_hip and self.dim_head in (32, 64) and self.heads <= 8 and self.kernel_size[0] > 1 and \
            not (self.training and self.dropout.p > 0)

    def _params(self):
        h, inner = self.heads, self.heads * self.dim_head
        w = self.to_qkv.weight
        taps = self.rel_pos_bias().reshape(self.kernel_size[0], h)
        bias = torch.cat((taps.new_zeros(1, h), taps), 0).float()
        return (w[:inner], w[inner:], self.talking_heads.weight.reshape(h, h, 1, 1), self.to_out.weight,
                w.new_zeros(self.to_out.weight.shape[0]), bias)

    def _meta(self, B, n, device, **_):
        from . import kernels as K
        g = K.s3_geom(B, n, (max(n - 1, 1), 1, 1), (self.kernel_size[0], 1, 1), (self.dilation[0], 1, 1), self.heads, self.dim_head)
        return dict(kind='s3', cache=self._cache, geom=g)

    def _forward_hip(self, x):
        return ops.InnerFn.apply(x, None, self._meta(x.shape[0], x.shape[1], x.device), *self._params())

    def forward(self, x, **kwargs):
        b, n, h = x.shape[0], x.shape[1], self.heads
        if self.training and self.dropout.p > 0:
            raise NotImplementedError('attention dropout inside SparseCausal2DNA is not built')
        if x.is_cuda and self._hip_ok():
            return self._forward_hip(x)
        q, k, v = self.to_qkv(x).chunk(3, dim=-1)
        if n == 1:
            return self.to_out(v)
        split = lambda t: t.reshape(b, n, h, -1).permute(0, 2, 1, 3)            # b h n d
        q, k, v = split(q) * self.scale, split(k), split(v)
        m = n - 1
        idx = self._taps(m, x.device)                                            # [m, kk]
        ok = idx >= 0
        gidx = idx.clamp(min=0) + 1                                              # token rows (1-based: row 0 is <bos>)
        kg, vg = k[:, :, gidx], v[:, :, gidx]                                    # b h m kk d
        qa = q[:, :, 1:]
        sim = einsum('b h i d, b h i j d -> b h i j', qa, kg)
        sim = sim + self.rel_pos_bias().reshape(-1, h).t()[None, :, None, :]     # [kk, h] -> per (head, tap)
        sim = sim.masked_fill(~ok[None, None], NEG)
        sim0 = einsum('b h i d, b h d -> b h i', qa, k[:, :, 0])[..., None]      # <bos> key, no bias, never masked
        attn = torch.cat((sim0, sim), dim=-1).softmax(dim=-1, dtype=torch.float32)
        attn = einsum('g h, b h i j -> b g i j', self.talking_heads.weight.reshape(h, h), attn)
        out = einsum('b h i j, b h i j d -> b h i d', attn[..., 1:], vg) + attn[..., :1] * v[:, :, :1]
        out = torch.cat((v[:, :, :1], out), dim=2)                               # <bos> row: its own value
        return self.to_out(out.permute(0, 2, 1, 3).reshape(b, n, -1))


class CrossModalityCrossAttention(nn.Module):
    """np.py:908-1067.  The sequence (minus its start token) is cut into frames of `chunk_size` tokens; the context is
    [0] * (context_chunk_size - 1) + [context start token] + context tokens, cut into frames of `context_chunk_size`: frame t
    of the sequence attends a learned null key/value + context frame t, i.e. the context one frame EARLIER in time (frame 0
    sees the context start token and cc - 1 zero rows, which do take softmax mass when no mask is given: quirk Q21).
    Talking heads here is a Conv3d WITH bias, applied after the softmax.  The start-token row of the output is 0."""

    def __init__(self, *, dim, chunk_size, context_chunk_size, heads=8, dim_head=64, context_dim=None, has_start_token=True,
                 context_has_start_token=True, norm=False, norm_context=False, dropout=0.):
        super().__init__()
        context_dim = default(context_dim, dim)
        inner = heads * dim_head
        self.heads, self.scale = heads, dim_head ** -0.5
        self.norm = nn.LayerNorm(dim) if norm else nn.Identity()
        self.context_norm = nn.LayerNorm(context_dim) if norm_context else nn.Identity()
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(context_dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)
        self.null_k = nn.Parameter(torch.randn(heads, dim_head))
        self.null_v = nn.Parameter(torch.randn(heads, dim_head))
        self.talking_heads = nn.Conv3d(heads, heads, 1)
        self.dropout = nn.Dropout(dropout)
        self.has_start_token, self.context_has_start_token = has_start_token, context_has_start_token
        self.chunk_size, self.context_chunk_size = chunk_size, context_chunk_size
        self.dim_head = dim_head
        self._cache = ops.WeightCache()
        self.use_hip = True            # False: the torch-op formulation (kept for head sizes the kernels do not cover)

    def forward(self, seq, context, mask=None, context_mask=None):
        if self.training and self.dropout.p > 0:
            raise NotImplementedError('attention dropout inside CrossModalityCrossAttention is not built')
        b, n_in, dim = seq.shape
        c, cc, h = self.chunk_size, self.context_chunk_size, self.heads
        body = seq[:, 1:] if self.has_start_token else seq
        s_len = body.shape[1]
        ns = -(-s_len // c)
        body = F.pad(body, (0, 0, 0, ns * c - s_len))
        c_len = context.shape[1] - (1 if self.context_has_start_token else 0)
        lead = cc - 1 if cc else 0
        tail = -(-c_len // cc) * cc - c_len
        ctx = F.pad(context, (0, 0, lead, tail))
        cmask = F.pad(context_mask, (lead, tail), value=False) if exists(context_mask) else None
        nc = ctx.shape[1] // cc
        nf = min(ns, nc)                                      # frames that have a context frame to look at
        if nf == 0:
            return torch.zeros_like(seq)
        qf = self.norm(body[:, :nf * c].reshape(b, nf, c, dim))
        cf = self.context_norm(ctx[:, :nf * cc].reshape(b, nf, cc, -1))
        if self.use_hip and seq.is_cuda and self.dim_head in (32, 64) and h <= 8 and cc + 1 <= 288:
            # every (sample, frame) pair is one sample of the libamdnuwa cross-attention kernels (null k/v, key mask, talking heads);
            # the Conv3d BIAS adds bias[g] to every attention weight, i.e. bias[g] * (null_v[g] + sum_j v_j[g]) to the head output
            from . import kernels as K
            inner = h * self.dim_head
            xq, xc = qf.reshape(b * nf, c, dim), cf.reshape(b * nf, cc, -1)
            g = K.x_geom(b * nf, c, cc, h, self.dim_head)
            km = cmask[:, :nf * cc].reshape(b * nf, cc).to(torch.uint8).contiguous() if exists(cmask) else None
            meta = dict(kind='xattn', cache=self._cache, xgeom=g, mask_u8=km, save=torch.is_grad_enabled())
            out = ops.InnerFn.apply(xq, xc, meta, self.null_k.reshape(h, 1, -1), self.null_v.reshape(h, 1, -1),
                                    self.talking_heads.weight.reshape(h, h, 1, 1), self.to_q.weight, self.to_kv.weight, self.to_out.weight)
            vsum = self.null_v[None] + F.linear(xc.sum(1), self.to_kv.weight[inner:]).reshape(b * nf, h, -1)
            corr = F.linear((self.talking_heads.bias[None, :, None] * vsum).reshape(b * nf, inner), self.to_out.weight)
            out = (out + corr[:, None]).reshape(b, nf * c, -1)
            out = F.pad(out, (0, 0, 0, max(0, s_len - nf * c)))[:, :s_len]
            if self.has_start_token:
                out = F.pad(out, (0, 0, 1, 0))
            if exists(mask):
                out = out.masked_fill(~mask[..., None], 0.)
            return out
        q = self.to_q(qf).reshape(b, nf, c, h, -1).permute(0, 3, 1, 2, 4) * self.scale          # b h f c d
        k, v = (t.reshape(b, nf, cc, h, -1).permute(0, 3, 1, 2, 4) for t in self.to_kv(cf).chunk(2, dim=-1))
        nk = self.null_k[None, :, None, None, :].expand(b, h, nf, 1, -1)
        nv = self.null_v[None, :, None, None, :].expand(b, h, nf, 1, -1)
        k, v = torch.cat((nk, k), dim=3), torch.cat((nv, v), dim=3)
        sim = einsum('b h f i d, b h f j d -> b h f i j', q, k)
        if exists(cmask):
            km = F.pad(cmask[:, :nf * cc].reshape(b, nf, cc), (1, 0), value=True)                # the null key is always visible
            sim = sim.masked_fill(~km[:, None, :, None, :], NEG)
        attn = sim.softmax(dim=-1, dtype=torch.float32)
        attn = einsum('g h, b h f i j -> b g f i j', self.talking_heads.weight.reshape(h, h), attn) + \
            self.talking_heads.bias[None, :, None, None, None]
        out = einsum('b h f i j, b h f j d -> b h f i d', attn, v)
        out = self.to_out(out.permute(0, 2, 3, 1, 4).reshape(b, nf * c, -1))
        out = F.pad(out, (0, 0, 0, max(0, s_len - nf * c)))[:, :s_len]       # frames without context output 0; drop the padding
        if self.has_start_token:
            out = F.pad(out, (0, 0, 1, 0))
        if exists(mask):
            out = out.masked_fill(~mask[..., None], 0.)
        return out


def _residual(block, x, **kw):
    """x + block(x) for a SandwichNorm block: the fused libamdnuwa node when its inner module is one of the hot ones"""
    inner_kw = {k: v for k, v in kw.items() if k in ('context', 'context_mask')}
    if isinstance(block, SandwichNorm) and x.is_cuda and block._inner(inner_kw.get('context')) is not None:
        return block.fused_residual(x, **inner_kw)
    return block(x, **{k: v for k, v in kw.items() if v is not None or k == 'context'}) + x


class DualModalityDecoder(nn.Module):
    """np.py:1299-1487: per depth a video triple (3DNA, text cross-attention, FF) and an audio triple (1-D causal window,
    text cross-attention, FF); after every `cross_modality_attn_every`-th depth a pair of cross-modality blocks."""

    def __init__(self, *, dim, depth, num_audio_tokens_per_video_frame, num_video_tokens_per_frame, sparse_3dna_video_shape, heads=8,
                 dim_head=64, ff_mult=4, attn_dropout=0., ff_dropout=0., ff_chunk_size=None, sparse_3dna_kernel_size=3,
                 sparse_3dna_query_num_frames_chunk=None, sparse_3dna_dilations=(1,), sparse_3dna_rel_pos_bias=False,
                 sparse_2dna_kernel_size=7, sparse_2dna_dilation=(1,), sparse_2dna_rel_pos_bias=False, shift_video_tokens=False,
                 shift_audio_tokens=False, audio_tokens_per_timestep=1, cross_modality_attn_every=3):
        super().__init__()
        self.layers = MList([])
        self.layer_types = []
        sn = lambda fn: SandwichNorm(dim=dim, fn=fn)
        ff = lambda: FeedForward(dim=dim, mult=ff_mult, dropout=ff_dropout, chunk_size=ff_chunk_size)
        text_attn = lambda: Attention(dim=dim, heads=heads, dim_head=dim_head, dropout=attn_dropout)
        fmap = sparse_3dna_video_shape[-1]
        vshift = (lambda fn: ShiftVideoTokens(fn, image_size=fmap)) if shift_video_tokens else (lambda fn: fn)
        ashift = (lambda fn: ShiftAudioTokens(fn, audio_tokens_per_timestep=audio_tokens_per_timestep)) if shift_audio_tokens else (lambda fn: fn)
        for ind in range(depth):
            video_attn = Sparse3DNA(dim=dim, heads=heads, dim_head=dim_head, causal=True, kernel_size=sparse_3dna_kernel_size,
                                    dilation=sparse_3dna_dilations[ind % len(sparse_3dna_dilations)],
                                    video_shape=sparse_3dna_video_shape, query_num_frames_chunk=sparse_3dna_query_num_frames_chunk,
                                    rel_pos_bias=sparse_3dna_rel_pos_bias)
            audio_attn = SparseCausal2DNA(dim=dim, heads=heads, dim_head=dim_head, kernel_size=sparse_2dna_kernel_size,
                                          dilation=sparse_2dna_dilation[ind % len(sparse_2dna_dilation)], dropout=attn_dropout,
                                          rel_pos_bias=sparse_2dna_rel_pos_bias)
            self.layer_types.append('intra_modality')
            self.layers.append(MList([MList([sn(vshift(video_attn)), sn(text_attn()), sn(vshift(ff()))]),
                                      MList([sn(ashift(audio_attn)), sn(text_attn()), sn(ashift(ff()))])]))
            if (ind + 1) % cross_modality_attn_every == 0:
                cross = lambda cs, ccs: MList([sn(CrossModalityCrossAttention(dim=dim, heads=heads, dim_head=dim_head, chunk_size=cs,
                                                                              context_chunk_size=ccs, has_start_token=True,
                                                                              context_has_start_token=True)), sn(ff())])
                self.layer_types.append('inter_modality')
                self.layers.append(MList([cross(num_video_tokens_per_frame, num_audio_tokens_per_video_frame),
                                          cross(num_audio_tokens_per_video_frame, num_video_tokens_per_frame)]))
        self.video_norm = StableLayerNorm(dim)
        self.audio_norm = StableLayerNorm(dim)

    def forward_layers(self, video, audio, *, context, audio_mask=None, video_mask=None, context_mask=None, **kwargs):
        for blocks, kind in zip(self.layers, self.layer_types):
            if kind == 'intra_modality':
                (v_attn, v_cross, v_ff), (a_attn, a_cross, a_ff) = blocks
                v = _residual(v_attn, video, mask=video_mask)
                v = _residual(v_cross, v, context=context, mask=video_mask, context_mask=context_mask)
                v = _residual(v_ff, v)
                a = _residual(a_attn, audio, mask=audio_mask)
                a = _residual(a_cross, a, context=context, mask=audio_mask, context_mask=context_mask)
                a = _residual(a_ff, a)
            else:
                (v_x, v_ff), (a_x, a_ff) = blocks
                v = v_x(video, context=audio, mask=video_mask, context_mask=audio_mask) + video     # both read the OLD streams
                a = a_x(audio, context=video, mask=audio_mask, context_mask=video_mask) + audio
                v, a = _residual(v_ff, v), _residual(a_ff, a)
            video, audio = v, a
        return video, audio

    def forward(self, video, audio, **kwargs):
        video, audio = self.forward_layers(video, audio, **kwargs)
        return self.video_norm(video), self.audio_norm(audio)


class DualReversibleBlock(nn.Module):
    """one block of the reversible dual decoder (reversible_video_audio.py:27-327): video halves (x1, x2) and audio halves
    (m1, m2); f / g act on video, j / k on audio -- except in the cross-modality block, where the reference feeds the video
    stream through `k` and the audio stream through `g` (reversible_video_audio.py:241-244; reproduced)."""

    def __init__(self, kind, f, g, j, k):
        super().__init__()
        self.kind = kind
        self.f, self.g, self.j, self.k = (Deterministic(t) for t in (f, g, j, k))

    def forward(self, x1, x2, m1, m2, *, context, context_mask, video_mask=None, audio_mask=None):
        f, g, j, k = self.f.net, self.g.net, self.j.net, self.k.net
        if self.kind == 'intra_modality_self_attn':
            y1 = _residual_to(f, x2, x1, mask=video_mask)
            y2 = _residual_to(g, y1, x2)
            n1 = _residual_to(j, m2, m1, mask=audio_mask)
            n2 = _residual_to(k, n1, m2)
        elif self.kind == 'intra_modality_cross_attn':
            y1 = _residual_to(f, x2, x1, context=context, context_mask=context_mask, mask=video_mask)
            y2 = _residual_to(g, y1, x2)
            n1 = _residual_to(j, m2, m1, context=context, context_mask=context_mask, mask=audio_mask)
            n2 = _residual_to(k, n1, m2)
        else:
            y1 = x1 + f(x2, m2, mask=video_mask, context_mask=audio_mask)
            y2 = _residual_to(k, y1, x2)
            n1 = m1 + j(m2, y2, mask=audio_mask, context_mask=video_mask)       # the audio side sees the UPDATED video half
            n2 = _residual_to(g, n1, m2)
        return y1, y2, n1, n2

    def backward_pass(self, y1, y2, n1, n2, dy1, dy2, dn1, dn2, *, context, context_mask, video_mask=None, audio_mask=None):
        """inputs and input-gradients of this block from its outputs and output-gradients; parameter (and text-context) gradients
        accumulate through the inner autograd calls"""
        f, g, j, k = self.f.net, self.g.net, self.j.net, self.k.net
        leaf = lambda t: t.detach().requires_grad_(True)
        gr = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)      # (a sub-block may ignore an input, e.g. no context frame yet)
        if self.kind != 'inter_modality_cross_attn':
            ckw = dict(context=context, context_mask=context_mask) if self.kind == 'intra_modality_cross_attn' else {}
            # video halves: y2 = x2 + g(y1), y1 = x1 + f(x2)
            # (add = the gradient the step's input gradient is summed with: on the fused node it rides in as the pre-norm backward's accumulator)
            y1g = leaf(y1)
            x2, dx1 = _rev_step(g, y1g, y2, dy2, add=dy1)
            x2g = leaf(x2)
            x1, dx2 = _rev_step(f, x2g, y1, dx1, add=dy2, mask=video_mask, **ckw)
            # audio halves: n2 = m2 + k(n1), n1 = m1 + j(m2)
            n1g = leaf(n1)
            m2, dm1 = _rev_step(k, n1g, n2, dn2, add=dn1)
            m2g = leaf(m2)
            m1, dm2 = _rev_step(j, m2g, n1, dm1, add=dn2, mask=audio_mask, **ckw)
            return x1, x2, m1, m2, dx1, dx2, dm1, dm2
        # cross-modality block: y1 = x1 + f(x2, m2); y2 = x2 + k(y1); n1 = m1 + j(m2, y2); n2 = m2 + g(n1)
        n1g = leaf(n1)
        m2, dm1 = _rev_step(g, n1g, n2, dn2, add=dn1)
        m2g, y2g = leaf(m2), leaf(y2)
        m1 = _rev_step(j, m2g, n1, dm1, extra=(y2g,), mask=audio_mask, context_mask=video_mask)
        dm2 = dn2 + gr(m2g)
        dy2t = dy2 + gr(y2g)                              # the audio side looked at the updated video half
        y1g = leaf(y1)
        x2, dx1 = _rev_step(k, y1g, y2, dy2t, add=dy1)
        x2g, m2h = leaf(x2), leaf(m2)
        x1 = _rev_step(f, x2g, y1, dx1, extra=(m2h,), mask=video_mask, context_mask=audio_mask)
        dx2 = dy2t + gr(x2g)
        dm2 = dm2 + gr(m2h)
        return x1, x2, m1, m2, dx1, dx2, dm1, dm2


def _rev_step(block, xg, y, dy, extra=(), add=None, **kw):
    """One reversal step of a reversible half: y = x_prev + block(xg, *extra, **kw).  Returns x_prev (no graph) after
    back-propagating dy through a freshly recomputed block: gradients land in xg.grad, in the .grad of any `extra` / `context`
    leaf, and accumulate into the block's parameters (the role of reversible_video_audio.py:246-327).  add: also returns add + xg.grad.
    On the fused libamdnuwa
    node the recomputation and the subtraction are one pass: the node's value is y - block(xg) = x_prev (`minus`)."""
    inner_kw = {a: b for a, b in kw.items() if a in ('context', 'context_mask')}
    with torch.enable_grad():
        if not extra and isinstance(block, SandwichNorm) and xg.is_cuda and block._inner(inner_kw.get('context')) is not None:
            xp = block.fused_residual(xg, resid=y, minus=True, **inner_kw)
            if add is not None:
                xp.grad_fn.dx_add = add                  # (xg.grad = add + dL/dxg: ops.SandwichBlockFn.backward)
            torch.autograd.backward(xp, dy)
            return xp.detach() if add is None else (xp.detach(), xg.grad)
        out = block(xg, *extra, **{a: b for a, b in kw.items() if b is not None})
        torch.autograd.backward(out, dy)
        x_prev = y - out.detach()
    if add is None:
        return x_prev
    return x_prev, (add + xg.grad if xg.grad is not None else add)


def _residual_to(block, x, resid, **kw):
    """resid + block(x): the fused libamdnuwa node (residual taken from a different tensor) where the block allows it"""
    inner_kw = {a: b for a, b in kw.items() if a in ('context', 'context_mask')}
    if isinstance(block, SandwichNorm) and x.is_cuda and block._inner(inner_kw.get('context')) is not None:
        return block.fused_residual(x, resid=resid, **inner_kw)
    return resid + block(x, **{a: b for a, b in kw.items() if b is not None})


class _DualReversibleStackFn(torch.autograd.Function):
    """O(1)-activation-memory execution of the reversible dual decoder (the role of reversible_video_audio.py:329-355): the
    forward keeps only the four output halves; the backward walks the blocks in reverse, rebuilding each block's inputs from its
    outputs while back-propagating through one recomputed sub-block at a time.  The text context is an explicit input so that
    its gradient leaves this node once, summed over the blocks."""

    @staticmethod
    def forward(ctx, video, audio, context, seq, kw):
        x1 = x2 = video.detach()
        m1 = m2 = audio.detach()
        cdet = context.detach() if context is not None else None
        for block in seq.blocks:
            x1, x2, m1, m2 = block(x1, x2, m1, m2, context=cdet, **kw)
        ctx.seq, ctx.kw, ctx.cdet = seq, kw, cdet
        ctx.save_for_backward(x1, x2, m1, m2)
        return (x1 + x2) * 0.5, (m1 + m2) * 0.5

    @staticmethod
    def backward(ctx, dv, da):
        y1, y2, n1, n2 = ctx.saved_tensors
        dy1 = dy2 = dv * 0.5
        dn1 = dn2 = da * 0.5
        dctx = None
        for block in reversed(list(ctx.seq.blocks)):
            leaf = ctx.cdet.detach().requires_grad_(True) if (ctx.cdet is not None and block.kind == 'intra_modality_cross_attn') else None
            y1, y2, n1, n2, dy1, dy2, dn1, dn2 = block.backward_pass(y1, y2, n1, n2, dy1, dy2, dn1, dn2, context=leaf, **ctx.kw)
            if leaf is not None and leaf.grad is not None:
                dctx = leaf.grad if dctx is None else dctx + leaf.grad
        return dy1 + dy2, dn1 + dn2, dctx, None, None


class DualModalityReversibleSequence(nn.Module):
    """reversible_video_audio.py:367-407: both streams are duplicated, run through the blocks, and the two halves AVERAGED.
    In training the stack runs through _DualReversibleStackFn (activations of one sub-block at a time, as the reference's
    recomputing backward); `memory_efficient = False` keeps every activation in the autograd graph instead (same forward)."""

    memory_efficient = True

    def __init__(self, input_blocks, block_types):
        super().__init__()
        self.block_types = block_types
        self.blocks = nn.ModuleList([DualReversibleBlock(kind, *blk) for blk, kind in zip(input_blocks, block_types)])

    def forward(self, video, audio, *, context, context_mask=None, video_mask=None, audio_mask=None, reverse=True):
        if self.memory_efficient and reverse and torch.is_grad_enabled() and video.is_cuda and \
                (video.requires_grad or audio.requires_grad or any(p.requires_grad for p in self.parameters())):
            kw = dict(context_mask=context_mask, video_mask=video_mask, audio_mask=audio_mask)
            return _DualReversibleStackFn.apply(video, audio, context, self, kw)
        x1 = x2 = video
        m1 = m2 = audio
        for block in self.blocks:
            x1, x2, m1, m2 = block(x1, x2, m1, m2, context=context, context_mask=context_mask, video_mask=video_mask,
                                   audio_mask=audio_mask)
        return (x1 + x2) * 0.5, (m1 + m2) * 0.5


class ReversibleDualModalityDecoder(nn.Module):
    """np.py:1489-1655: per depth a self-attention block (video 3DNA + FF, audio window attention + FF), a text
    cross-attention block, and after every `cross_modality_attn_every`-th depth a cross-modality block whose attention and
    FeedForward modules are NOT SandwichNorm-wrapped."""

    def __init__(self, *, dim, depth, num_audio_tokens_per_video_frame, num_video_tokens_per_frame, sparse_3dna_video_shape, heads=8,
                 dim_head=64, ff_mult=4, attn_dropout=0., ff_dropout=0., ff_chunk_size=None, sparse_3dna_kernel_size=3,
                 sparse_3dna_query_num_frames_chunk=None, sparse_3dna_dilations=(1,), sparse_3dna_rel_pos_bias=False,
                 sparse_2dna_kernel_size=7, sparse_2dna_dilation=(1,), sparse_2dna_rel_pos_bias=False, shift_video_tokens=False,
                 shift_audio_tokens=False, audio_tokens_per_timestep=1, cross_modality_attn_every=3):
        super().__init__()
        self.layers = MList([])
        self.layer_types = []
        sn = lambda fn: SandwichNorm(dim=dim, fn=fn)
        ff = lambda: FeedForward(dim=dim, mult=ff_mult, dropout=ff_dropout, chunk_size=ff_chunk_size)
        text_attn = lambda: Attention(dim=dim, heads=heads, dim_head=dim_head, dropout=attn_dropout)
        fmap = sparse_3dna_video_shape[-1]
        vshift = (lambda fn: ShiftVideoTokens(fn, image_size=fmap)) if shift_video_tokens else (lambda fn: fn)
        ashift = (lambda fn: ShiftAudioTokens(fn, audio_tokens_per_timestep=audio_tokens_per_timestep)) if shift_audio_tokens else (lambda fn: fn)
        for ind in range(depth):
            video_attn = Sparse3DNA(dim=dim, heads=heads, dim_head=dim_head, causal=True, kernel_size=sparse_3dna_kernel_size,
                                    dilation=sparse_3dna_dilations[ind % len(sparse_3dna_dilations)],
                                    video_shape=sparse_3dna_video_shape, query_num_frames_chunk=sparse_3dna_query_num_frames_chunk,
                                    rel_pos_bias=sparse_3dna_rel_pos_bias)
            audio_attn = SparseCausal2DNA(dim=dim, heads=heads, dim_head=dim_head, dropout=attn_dropout,
                                          kernel_size=sparse_2dna_kernel_size,
                                          dilation=sparse_2dna_dilation[ind % len(sparse_2dna_dilation)], rel_pos_bias=sparse_2dna_rel_pos_bias)
            # (argument order of the four lambdas = construction order in the reference: attention modules first, then the FFs)
            v_ff, a_ff = ff(), ff()
            self.layers.append(MList([sn(vshift(video_attn)), sn(vshift(v_ff)), sn(ashift(audio_attn)), sn(ashift(a_ff))]))
            self.layer_types.append('intra_modality_self_attn')
            v_x, a_x = text_attn(), text_attn()
            self.layers.append(MList([sn(v_x), sn(ff()), sn(a_x), sn(ff())]))
            self.layer_types.append('intra_modality_cross_attn')
            if (ind + 1) % cross_modality_attn_every == 0:
                xm = lambda cs, ccs: CrossModalityCrossAttention(dim=dim, heads=heads, dim_head=dim_head, chunk_size=cs,
                                                                 context_chunk_size=ccs, has_start_token=True, context_has_start_token=True)
                v2a = xm(num_video_tokens_per_frame, num_audio_tokens_per_video_frame)
                v_xff = ff()
                a2v = xm(num_audio_tokens_per_video_frame, num_video_tokens_per_frame)
                a_xff = ff()
                self.layers.append(MList([v2a, v_xff, a2v, a_xff]))
                self.layer_types.append('inter_modality_cross_attn')
        self.net = DualModalityReversibleSequence(self.layers, self.layer_types)
        self.video_norm = StableLayerNorm(dim)
        self.audio_norm = StableLayerNorm(dim)
        for m in self.modules():                 # (see ReversibleTransformer: no kept dropout mask where the forward is recomputed)
            if isinstance(m, FeedForward):
                m._no_hip_dropout = True

    def forward_layers(self, video, audio, *, context, audio_mask=None, video_mask=None, context_mask=None, **kwargs):
        return self.net(video, audio, context=context, audio_mask=audio_mask, video_mask=video_mask, context_mask=context_mask)

    def forward(self, video, audio, **kwargs):
        video, audio = self.forward_layers(video, audio, **kwargs)
        return self.video_norm(video), self.audio_norm(audio)


class NUWAVideoAudio(nn.Module):
    """np.py:1968-2293: identical constructor kwargs and forward() signature (training loss / logits)."""

    generate_use_cache = True          # key/value-cached generate() (decode.DualGuidedStepper; plain and reversible dual decoder)
    generate_use_graph = True          # ... with the ordinary rows of each stream replayed from a captured HIP graph

    def __init__(self, *, vae, dim, image_size, num_audio_tokens, num_audio_tokens_per_video_frame, audio_tokens_per_timestep=1,
                 max_video_frames=5, text_num_tokens=49408, text_max_seq_len=256, text_enc_depth=6, text_enc_dim_head=64,
                 text_enc_heads=8, text_rotary_pos_emb=False, enc_reversible=False, dec_reversible=True, dec_depth=6, dec_dim_head=64,
                 dec_heads=8, attn_dropout=0., ff_dropout=0., ff_chunk_size=None, embed_gradient_frac=0.2, shift_video_tokens=True,
                 shift_audio_tokens=True, sparse_3dna_kernel_size=3, sparse_3dna_query_num_frames_chunk=None, sparse_3dna_dilation=1,
                 sparse_3dna_rel_pos_bias=True, sparse_2dna_kernel_size=7, sparse_2dna_dilation=1, sparse_2dna_rel_pos_bias=True,
                 audio_loss_weight=1., cross_modality_attn_every=3):
        super().__init__()
        self.vae = vae.copy_for_eval()
        num_image_tokens = vae.codebook_size
        self.text_max_seq_len = text_max_seq_len
        self.text_embedding = Embedding(text_num_tokens, dim, frac_gradient=embed_gradient_frac)
        self.text_abs_pos_emb = Embedding(text_max_seq_len, dim) if not text_rotary_pos_emb else None
        self.text_rotary_pos_emb = RotaryEmbedding(dim=min(32, text_enc_dim_head)) if text_rotary_pos_emb else None
        enc_klass = ReversibleTransformer if enc_reversible else Transformer
        self.text_transformer = enc_klass(dim=dim, depth=text_enc_depth, heads=text_enc_heads, dim_head=text_enc_dim_head,
                                          attn_dropout=attn_dropout, ff_dropout=ff_dropout)
        self.video_bos = nn.Parameter(torch.randn(dim))
        self.image_embedding = Embedding(num_image_tokens, dim, frac_gradient=embed_gradient_frac)
        fmap_size = image_size // (2 ** vae.num_layers)
        self.video_fmap_size, self.max_video_frames = fmap_size, max_video_frames
        self.video_shape = (max_video_frames, fmap_size, fmap_size)
        self.video_pos_emb = AxialPositionalEmbedding(dim, shape=self.video_shape)
        self.audio_bos = nn.Parameter(torch.randn(dim))
        self.audio_embedding = Embedding(num_audio_tokens, dim, frac_gradient=embed_gradient_frac)
        # (the reference sizes this table by the audio CODEBOOK size, not by the sequence length)
        self.audio_pos_emb = AxialPositionalEmbedding(dim, shape=(num_audio_tokens // audio_tokens_per_timestep, audio_tokens_per_timestep))
        self.audio_loss_weight = audio_loss_weight
        self.num_video_tokens_per_frame = fmap_size ** 2
        self.num_audio_tokens_per_video_frame = num_audio_tokens_per_video_frame
        as_cycle = lambda d: tuple(d) if isinstance(d, (list, tuple)) else tuple(range(1, d + 1))
        decoder_klass = ReversibleDualModalityDecoder if dec_reversible else DualModalityDecoder
        self.video_audio_transformer = decoder_klass(
            dim=dim, depth=dec_depth, heads=dec_heads, dim_head=dec_dim_head, attn_dropout=attn_dropout, ff_dropout=ff_dropout,
            ff_chunk_size=ff_chunk_size, audio_tokens_per_timestep=audio_tokens_per_timestep, shift_audio_tokens=shift_audio_tokens,
            shift_video_tokens=shift_video_tokens, sparse_3dna_video_shape=self.video_shape,
            sparse_3dna_kernel_size=sparse_3dna_kernel_size, sparse_3dna_dilations=as_cycle(sparse_3dna_dilation),
            sparse_3dna_query_num_frames_chunk=sparse_3dna_query_num_frames_chunk, sparse_3dna_rel_pos_bias=sparse_3dna_rel_pos_bias,
            num_audio_tokens_per_video_frame=num_audio_tokens_per_video_frame, num_video_tokens_per_frame=fmap_size * fmap_size,
            cross_modality_attn_every=cross_modality_attn_every, sparse_2dna_kernel_size=sparse_2dna_kernel_size,
            sparse_2dna_dilation=as_cycle(sparse_2dna_dilation), sparse_2dna_rel_pos_bias=sparse_2dna_rel_pos_bias)
        self.to_video_logits = nn.Linear(dim, num_image_tokens, bias=False)
        self.to_audio_logits = nn.Linear(dim, num_audio_tokens, bias=False)
        self._cache_v, self._cache_a = ops.WeightCache(), ops.WeightCache()

    def embed_text(self, text, mask=None):
        batch, seq_len, device = *text.shape, text.device
        assert seq_len <= self.text_max_seq_len, 'your input text has a greater length than what was designated on initialization'
        tokens = self.text_embedding(text)
        if exists(self.text_abs_pos_emb):
            tokens = tokens + self.text_abs_pos_emb(torch.arange(seq_len, device=device))[None]
        rotary = self.text_rotary_pos_emb(seq_len, device=device) if exists(self.text_rotary_pos_emb) else None
        return self.text_transformer(tokens, mask=mask, rotary_pos_emb=rotary)

    def embed_video(self, ids_in):
        pe = self.video_pos_emb
        if ids_in.is_cuda and pe.num_axials == 3:
            frac = self.image_embedding.frac_gradient if self.training else 1.
            return ops.EmbedAssembleFn.apply(ids_in, self.image_embedding.embed.weight, pe.axial1, pe.axial2, pe.axial3,
                                             self.video_bos, self.video_shape, float(frac))
        emb = pe()[:ids_in.shape[1]] + self.image_embedding(ids_in)
        return torch.cat((self.video_bos[None, None].expand(ids_in.shape[0], 1, -1), emb), dim=1)

    def embed_audio(self, ids_in):
        emb = self.audio_embedding(ids_in)
        emb = emb + self.audio_pos_emb()[:emb.shape[1]][None]
        return torch.cat((self.audio_bos[None, None].expand(ids_in.shape[0], 1, -1), emb), dim=1)

    @torch.no_grad()
    def generate(self, *, text, filter_thres=0.9, temperature=1., decode_max_batchsize=10, cond_scale=2., num_frames=None):
        """np.py:2111-2222: video and audio tokens are sampled alternately, one video frame's worth at a time; every step runs the
        dual decoder over the whole prefix (twice with classifier-free guidance, the second pass fed the first pass's normed
        OUTPUTS as the reference does).  Returns (video [b, f, 3, H, W], audio token ids [b, f * audio_tokens_per_frame])."""
        was_training = self.training
        self.eval()
        try:
            return self._generate(text, filter_thres, temperature, decode_max_batchsize, cond_scale, num_frames)
        finally:
            self.train(was_training)

    def _generate(self, text, filter_thres, temperature, decode_max_batchsize, cond_scale, num_frames):
        from .nuwa_pytorch import map_in_chunks, sample_top_fraction
        if not text.is_cuda:
            raise RuntimeError('nuwa_pytorch_amd: the decoder path needs a HIP device; there is no CPU fallback')
        batch, device = text.shape[0], text.device
        tpf, apf = self.num_video_tokens_per_frame, self.num_audio_tokens_per_video_frame
        text_mask = text != 0
        text_embeds = self.embed_text(text, mask=text_mask)
        video_indices = torch.empty((batch, 0), device=device, dtype=torch.long)
        audio_indices = torch.empty((batch, 0), device=device, dtype=torch.long)
        num_frames = default(num_frames, self.max_video_frames)
        total_video_tokens, total_audio_tokens = num_frames * tpf, num_frames * apf
        dec = self.video_audio_transformer
        if self.generate_use_cache and num_frames <= self.max_video_frames:
            try:
                from .decode import DualGuidedStepper
                stepper = DualGuidedStepper(self, text_embeds, text_mask, total_video_tokens + 1, total_audio_tokens + 1, cond_scale,
                                            graph=self.generate_use_graph)
            except NotImplementedError:
                stepper = None                  # a block outside the single-row kernels: the recompute loop below
            if stepper is not None:
                return self._generate_cached(stepper, batch, total_video_tokens, total_audio_tokens, filter_thres, temperature,
                                             decode_max_batchsize)
        vn, an = dec.video_norm.norm, dec.audio_norm.norm
        no_text = torch.zeros_like(text_mask).bool()
        decoding_video = True
        while video_indices.shape[1] < total_video_tokens or audio_indices.shape[1] < total_audio_tokens:
            video_in = video_indices
            if video_indices.shape[1] > total_video_tokens:           # (np.py:2149-2153; never true inside this loop)
                curr = video_indices.shape[1] % tpf
                video_in = video_indices[:, -((self.max_video_frames - (0 if curr == 0 else 1)) * tpf + curr):]
            frame_emb = self.embed_video(video_in) if video_in.shape[1] > 0 else \
                self.video_bos[None, None].expand(batch, 1, -1).contiguous()
            audio_emb = self.embed_audio(audio_indices).contiguous()

            def logits_of(v_hid, a_hid):
                if decoding_video:
                    return ops.LogitsFn.apply(v_hid[:, -1:].contiguous(), vn.weight, vn.bias, self.to_video_logits.weight, self._cache_v)
                return ops.LogitsFn.apply(a_hid[:, -1:].contiguous(), an.weight, an.bias, self.to_audio_logits.weight, self._cache_a)
            v_hid, a_hid = dec.forward_layers(frame_emb, audio_emb, context=text_embeds, context_mask=text_mask)
            logits = logits_of(v_hid, a_hid)
            if cond_scale != 1:
                uv, ua = dec.forward_layers(dec.video_norm(v_hid), dec.audio_norm(a_hid), context=text_embeds, context_mask=no_text)
                uncond = logits_of(uv, ua)
                logits = uncond + (logits - uncond) * cond_scale
            sample = sample_top_fraction(logits[:, -1], filter_thres, temperature)[:, None]
            if decoding_video:
                video_indices = torch.cat((video_indices, sample), dim=1)
                boundary = video_indices.shape[1] % tpf == 0
            else:
                audio_indices = torch.cat((audio_indices, sample), dim=1)
                boundary = audio_indices.shape[1] % apf == 0
            if boundary:                                              # alternate, one video frame at a time
                decoding_video = not decoding_video
        self.last_generated_ids = video_indices
        return self._frames_from_ids(video_indices, batch, decode_max_batchsize), audio_indices

    def _frames_from_ids(self, video_indices, batch, decode_max_batchsize):
        from .nuwa_pytorch import map_in_chunks
        fs = self.video_fmap_size
        codes = self.vae.codes_for_decoder(video_indices)
        codes = codes.reshape(batch, -1, fs, fs, codes.shape[-1]).permute(0, 1, 4, 2, 3).reshape(-1, codes.shape[-1], fs, fs)
        images = map_in_chunks(codes.contiguous(), self.vae._hip_decode, chunks=decode_max_batchsize)
        return images.reshape(batch, -1, *images.shape[1:])

    def _generate_cached(self, stepper, batch, total_v, total_a, filter_thres, temperature, decode_max_batchsize):
        """the sampling order of np.py:2143-2207 with one new decoder row per sampled token (decode.DualGuidedStepper) instead of
        two passes over the whole prefix: a token's row is computed right after it is sampled, and its logits are kept until its
        stream is next asked for a token (rows never change once computed: see decode.py)"""
        from .nuwa_pytorch import sample_top_fraction
        tpf, apf = self.num_video_tokens_per_frame, self.num_audio_tokens_per_video_frame
        vpos, apos = self.video_pos_emb(), self.audio_pos_emb()
        ids = {'v': [], 'a': []}
        total = {'v': total_v, 'a': total_a}
        logits = {'v': stepper.advance('v', self.video_bos[None].expand(batch, -1)),
                  'a': stepper.advance('a', self.audio_bos[None].expand(batch, -1))}
        which = 'v'
        while len(ids['v']) < total_v or len(ids['a']) < total_a:
            token = sample_top_fraction(logits[which], filter_thres, temperature)
            t = len(ids[which])
            ids[which].append(token)
            if t + 1 < total[which]:
                row = (self.image_embedding(token) + vpos[t]) if which == 'v' else (self.audio_embedding(token) + apos[t])
                logits[which] = stepper.advance(which, row)
            if (t + 1) % (tpf if which == 'v' else apf) == 0:         # alternate, one video frame at a time
                which = 'a' if which == 'v' else 'v'
        video_indices, audio_indices = torch.stack(ids['v'], 1), torch.stack(ids['a'], 1)
        self.last_generated_ids = video_indices            # the token ids behind the returned frames
        return self._frames_from_ids(video_indices, batch, decode_max_batchsize), audio_indices

    def forward(self, *, text, video, audio, return_loss=False, cond_dropout_prob=0.2):
        batch, device = text.shape[0], text.device
        if not text.is_cuda:
            raise RuntimeError('nuwa_pytorch_amd: the decoder path needs a HIP device; there is no CPU fallback')
        text_mask = text != 0
        text_embeds = self.embed_text(text, mask=text_mask)
        if video.dtype == torch.long:
            frame_indices = video
        else:
            assert video.shape[1] == self.max_video_frames, f'you must give the full video frames ({self.max_video_frames}) during training'
            frame_indices = self.vae.get_video_indices(video)
        frame_indices = frame_indices.reshape(batch, -1)
        frame_emb = self.embed_video(frame_indices[:, :-1] if return_loss else frame_indices)
        audio_emb = self.embed_audio(audio[:, :-1] if return_loss else audio)
        if self.training and cond_dropout_prob > 0:
            text_mask = text_mask & ~bernoulli_rows(batch, cond_dropout_prob, device)[:, None]
        dec = self.video_audio_transformer
        v_hid, a_hid = dec.forward_layers(frame_emb, audio_emb, context=text_embeds, context_mask=text_mask)
        vn, an = dec.video_norm.norm, dec.audio_norm.norm
        if not return_loss:
            return (ops.LogitsFn.apply(v_hid, vn.weight, vn.bias, self.to_video_logits.weight, self._cache_v),
                    ops.LogitsFn.apply(a_hid, an.weight, an.bias, self.to_audio_logits.weight, self._cache_a))
        video_loss = ops.LogitsLossFn.apply(v_hid, frame_indices, vn.weight, vn.bias, self.to_video_logits.weight, self._cache_v)
        audio_loss = ops.LogitsLossFn.apply(a_hid, audio, an.weight, an.bias, self.to_audio_logits.weight, self._cache_a)
        return video_loss + audio_loss * self.audio_loss_weight
