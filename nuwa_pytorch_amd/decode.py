"""Key/value-cached token-by-token decoding for NUWA.generate (reference np.py:1841-1915).

The reference recomputes the whole prefix -- twice, for classifier-free guidance -- for every sampled token.  Every stage of
the decoder is causal (Sparse3DNA taps, np.py:420-457; ShiftVideoTokens, np.py:210-253) or row-wise (cross-attention over the
text, FeedForward, the norms), so row `pos` of the decoder depends only on rows <= pos.  IncrementalDecoder keeps, per layer,
  * the pre-norm outputs of the two token-shifted blocks (the shift reads the rows one grid row up / one token left),
  * the Sparse3DNA key / value rows,
  * the packed text keys / values of the cross-attention (computed once),
and computes ONE new row per call with the same libamdnuwa kernels the training path uses (GEMMs, LayerNorms, GEGLU, the
cross-attention core with n = 1) plus the two single-row kernels of csrc/decode.hip.  The row index lives in device memory,
so the per-token work of a whole guided step can be captured once in a HIP graph and replayed for every token.

NUWAVideoAudio.generate (np.py:2111-2222) decodes two interleaved streams through the DualModalityDecoder (np.py:1299-1487).  Every
stage is row-causal there too: the audio window attention looks back only, and the chunked video <-> audio attention lets frame t
of one stream see frame t - 1 of the other (np.py:908-1067), which is complete -- and final -- by the time any row of frame t is
computed, because the sampler alternates one video frame / one audio frame.  DualIncrementalDecoder therefore keeps one
IncrementalDecoder per modality (own position counter) and links the two at the cross-modality layers: each stream stores its
layer-input rows projected with the OTHER direction's to_kv, and attends the other stream's stored rows of the matching frame."""
import torch
import torch.nn.functional as F

from . import kernels as K
from . import ops


class _Block:
    """one sub-block of a row program: out_slot <- out_slot + post(inner(pre(in_slot)));  pre / post = None for the un-normed modules
    of the reversible dual decoder's cross-modality layer"""
    __slots__ = ('kind', 'pre', 'post', 'inner', 'fmap', 'hcache', 'kvcache', 'geom', 'pk', 'xg', 'o_const', 'xm', 'src', 'dst',
                 'store_before', 'store_after', 'c2')


def _cast_row(x, lo):
    out = K.empty_bf(tuple(x.shape), x.device, lo=lo)
    K.cast_pad(x, out)
    return out


class IncrementalDecoder:
    """One decoder pass (conditioned or not) over a growing sequence.  transformer: nuwa_pytorch.Transformer (non-reversible)
    whose blocks are all on the fused HIP path; context [B, T, D] fp32 and context_mask [B, T] bool as in Transformer.forward.

    Instead of `transformer`, `block_list` gives the row program explicitly: dicts with `mod` (a SandwichNorm block, or a bare
    FeedForward / CrossModalityCrossAttention), optional `context` (text rows for a cross-attention), `xm` (_XmDirection this block
    attends through), `dst` (the residual half it adds into: reversible stacks keep two, np.py's reversible.py; the block reads the
    OTHER half), `store_before` / `store_after` (_XmDirection objects that take this row's input / output as context); `combine`
    scales the sum of the two halves at the end (1 for the ReversibleTransformer, which is detected and laid out here; 0.5 for the
    reversible dual decoder)."""

    def __init__(self, transformer, batch, max_rows, context, context_mask, pos_dev, block_list=None, halves=1, combine=0.5):
        from .nuwa_pytorch import Attention, FeedForward, Sparse3DNA, SandwichNorm, SparseCross2DNA
        from .video_audio import SparseCausal2DNA
        dev = context.device
        if block_list is None and hasattr(transformer, 'net'):         # ReversibleTransformer (np.py:1184-1295): (f, g) pairs, y1 = x1 +
            from .nuwa_pytorch import ShiftVideoTokens                  # f(x2), y2 = x2 + g(y1), output = the SUM of the halves
            block_list, halves, combine = [], 2, 1.0
            for f, g in transformer.layers:
                is_cross = not isinstance(f.fn, ShiftVideoTokens)
                block_list += [dict(mod=f, context=context if is_cross else None, dst=0), dict(mod=g, dst=1)]
        self.B, self.rows, self.pos_dev, self.halves, self.combine = batch, max_rows, pos_dev, halves, combine
        lo = self.lo = K.want_lo()
        D = context.shape[-1]
        ctx_bf = ops._ctx_to_bf(context)
        mask_u8 = context_mask.to(torch.uint8).contiguous() if context_mask is not None else None
        # the text-masked pass of classifier-free guidance: every query sees only the learned null key, so the attention
        # output is the same row for every position -- (sum_h W_th[g, h]) * null_v[g] -- and is computed once
        all_masked = context_mask is not None and not bool(context_mask.any())
        self.blocks = []
        if block_list is None:
            block_list = [dict(mod=sn, context=ctx_arg) for attn, cross, ff in transformer.layers
                          for sn, ctx_arg in ((attn, None), (cross, context), (ff, None)) if sn is not None]
        for spec in block_list:
            mod, ctx_arg = spec['mod'], spec.get('context')
            blk = _Block()
            blk.kvcache = blk.geom = blk.pk = blk.xg = blk.o_const = blk.hcache = blk.fmap = blk.c2 = None
            blk.xm = spec.get('xm')
            blk.dst = spec.get('dst', 0)
            blk.src = (1 - blk.dst) if halves == 2 else 0
            blk.store_before, blk.store_after = spec.get('store_before', ()), spec.get('store_after', ())
            if isinstance(mod, SandwichNorm):
                blk.pre = (mod.prenorm.weight.detach(), mod.prenorm.bias.detach())
                blk.post = (mod.postnorm.weight.detach(), mod.postnorm.bias.detach())
                if blk.xm is not None:
                    inner, fmap = mod.fn, None
                else:
                    found = mod._inner(ctx_arg, seq_len=max_rows)
                    if found is None and isinstance(mod.fn, SparseCausal2DNA) and mod.fn._hip_ok():
                        found = (mod.fn, None)                         # audio tower built without the channel shift
                    if found is None:
                        raise NotImplementedError('IncrementalDecoder: a decoder block is not on the libamdnuwa path')
                    inner, fmap = found
            else:                                                      # bare module: no norms, no shift
                blk.pre = blk.post = None
                inner, fmap = mod, None
                if not (blk.xm is not None or (isinstance(mod, FeedForward) and not mod._dropout_active())):
                    raise NotImplementedError(f'IncrementalDecoder: no single-row path for a bare {type(mod).__name__}')
            blk.inner, blk.fmap = inner, fmap
            blk.hcache = K.zeros_bf((batch, max_rows, D), dev, lo=lo) if fmap is not None else None
            if blk.xm is not None:
                blk.kind = 'xm'
            elif isinstance(inner, SparseCausal2DNA):               # the audio window attention IS a 3DNA over a (time, 1, 1) grid
                blk.kind = 's3'
                blk.geom = K.s3_geom(batch, max_rows, (max(max_rows - 1, 1), 1, 1), (inner.kernel_size[0], 1, 1),
                                     (inner.dilation[0], 1, 1), inner.heads, inner.dim_head)
                blk.kvcache = K.zeros_bf((batch, max_rows, 2 * inner.heads * inner.dim_head), dev, lo=lo)
            elif isinstance(inner, Sparse3DNA):
                if not inner.causal:
                    raise NotImplementedError('IncrementalDecoder needs causal Sparse3DNA')
                blk.kind = 's3'
                blk.geom = K.s3_geom(batch, max_rows, inner.video_shape, inner.kernel_size, inner.dilation, inner.heads,
                                     inner.dim_head)
                blk.kvcache = K.zeros_bf((batch, max_rows, 2 * inner.heads * inner.dim_head), dev, lo=lo)
            elif isinstance(inner, Attention):
                blk.kind = 'x'
                p = inner._params()
                W = ops.XInner.weights(inner._cache, p)
                blk.xg = K.x_geom(batch, 1, context.shape[1], inner.heads, inner.dim_head)
                kv = K.gemm_nt(ctx_bf, W['kv'], out_bf16=True)                 # text keys / values: once per sequence
                blk.pk = K.xattn_pack(blk.xg, kv, p[0].detach().reshape(inner.heads, inner.dim_head).contiguous(),
                                      p[1].detach().reshape(inner.heads, inner.dim_head).contiguous(), mask_u8)
                if all_masked:
                    q0 = K.zeros_bf((batch, inner.heads * inner.dim_head), dev, lo=lo)
                    blk.o_const = K.xattn_decode(blk.xg, q0, blk.pk, p[2].detach().reshape(inner.heads, inner.heads).contiguous())
            elif isinstance(inner, SparseCross2DNA):
                blk.kind = 'xc2'
                blk.c2 = _Cross2DNARows(inner, batch, ctx_bf, mask_u8, all_masked, lo)
            elif isinstance(inner, FeedForward):
                blk.kind = 'ff'
            else:
                raise NotImplementedError(f'IncrementalDecoder: no single-row path for {type(inner).__name__}')
            self.blocks.append(blk)
        self.bos_row_differs = any(b.kind == 'xc2' for b in self.blocks)       # row 0 takes another code path: not one graph for all rows

    def _enter(self, x, nxt):
        """the operand row of block `nxt` from its fp32 input row: pre-norm (+ token shift through the block's cache), or a plain
        cast for an un-normed block"""
        if nxt is None:
            return None
        if nxt.pre is None:
            return _cast_row(x, self.lo)
        return K.decode_ln(x, None, None, nxt.pre, cache=nxt.hcache, pos_dev=self.pos_dev, fmap=nxt.fmap or 0)[1]

    def step(self, x, bos=False):
        """x fp32 [B, D]: decoder input row `pos` of every sample -> that row after all layers (before the final norm).
        bos: this is row 0 (only a SparseCross2DNA block cares: its <bos> query attends to the whole context; the position itself
        lives in device memory and is not read on the host)"""
        fast = ops._fast()
        blocks = self.blocks
        state = [x] * self.halves
        h = self._enter(x, blocks[0])
        for i, blk in enumerate(blocks):
            inner = blk.inner
            for d in blk.store_before:
                d.store(state[blk.src], self.pos_dev)
            raw = blk.post is None                  # (an un-normed block adds its fp32 output itself)
            if blk.kind == 's3':
                p = inner._params()
                W = ops.S3Inner.weights(inner._cache, p)
                g = blk.geom
                rel = p[5].detach().contiguous() if len(p) > 5 else None
                qkv = K.gemm_nt(h, W['qkv'], out_bf16=True)
                o = K.s3_decode(g, qkv, blk.kvcache, self.pos_dev, p[2].detach().reshape(g.heads, g.heads).contiguous(), rel)
                y = K.gemm_nt(o, W['out'], bias=p[4].detach(), out_bf16=fast and not raw)
            elif blk.kind == 'xm':
                y = blk.xm.attend(h)
            elif blk.kind == 'x':
                p = inner._params()
                W = ops.XInner.weights(inner._cache, p)
                g = blk.xg
                wth = p[2].detach().reshape(g.heads, g.heads).contiguous()
                if blk.o_const is not None:
                    o = blk.o_const
                else:
                    q = K.gemm_nt(h, W['q'], out_bf16=True)
                    o = K.xattn_decode(g, q, blk.pk, wth)
                y = K.gemm_nt(o, W['out'], out_bf16=fast and not raw)
            elif blk.kind == 'xc2':
                W = ops.XInner.weights(inner._cache, inner._params())
                o = blk.c2.attend(h, W, self.pos_dev, bos)
                y = K.gemm_nt(o, W['out'], out_bf16=fast and not raw)
            else:
                W = ops.FFInner.weights(inner._cache, inner._params())
                u = K.gemm_nt(h, W['w1'], out_bf16=True)
                gg = K.geglu_fwd(u, W['FP'], interleaved=True)
                y = K.gemm_nt(gg, W['w2'], out_bf16=fast and not raw)
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            if not raw:
                # post-norm + residual, the next block's pre-norm and its token shift (cache write + gather): one launch
                fused = nxt is not None and nxt.pre is not None
                x, h = K.decode_ln(y, state[blk.dst], blk.post, nxt.pre if fused else None, cache=nxt.hcache if fused else None,
                                   pos_dev=self.pos_dev, fmap=(nxt.fmap or 0) if fused else 0)
                if nxt is not None and not fused:
                    h = _cast_row(x, self.lo)
            else:
                x = state[blk.dst] + y
                h = self._enter(x, nxt)
            state[blk.dst] = x
            for d in blk.store_after:
                d.store(x, self.pos_dev)
        return x if self.halves == 1 else (state[0] + state[1]) * self.combine


class _Cross2DNARows:
    """SparseCross2DNA (np.py:761-901) for one new query row per sample (NUWASketch.generate, np.py:2440-2512).  The block is row-wise:
    query row pos (> 0) sits at feature-map position i = (pos - 1) mod fmap^2 and attends to the learned null key + the kernel^2
    neighbourhood of i in EVERY sketch frame.  to_kv(context) is computed once; per row the window's key / value rows are gathered
    (index tables on the device, indexed by the device-side position, so the step stays capturable in a HIP graph), packed with
    amdnuwa_xattn_pack (padding slots and masked sketch tokens through its key mask) and attended with the single-query
    cross-attention kernel (fp32 softmax, talking heads).  Row 0 (<bos>) attends to ALL context tokens without talking heads
    (np.py:826-849): B rows of glue arithmetic, as in the training path (ops.XC2Inner).  With every context token masked (the
    second pass of classifier-free guidance) both outputs are constants, computed once."""

    def __init__(self, mod, batch, ctx_bf, mask_u8, all_masked, lo):
        dev = ctx_bf.hi.device
        self.mod, self.B, self.lo = mod, batch, lo
        h, dh = mod.heads, mod.dim_head
        self.inner = h * dh
        tpf = self.tpf = mod.image_size ** 2
        T = ctx_bf.hi.shape[0] // batch
        if T % tpf:
            raise NotImplementedError('cached decoding: the sketch context is not a whole number of frames')
        fs = T // tpf
        nbr = mod._nbr.to(dev)                                               # (tpf, k^2) in-frame neighbours, -1 = padding
        idx = torch.cat([torch.where(nbr >= 0, nbr + a * tpf, torch.zeros_like(nbr)) for a in range(fs)], dim=1)
        self.win_idx = idx.contiguous()                                      # (tpf, fs * k^2) context rows of every window slot
        self.win_ok = (nbr >= 0).repeat(1, fs).to(torch.uint8).contiguous()  # 0 = the slot is 'same' padding
        J = self.win_idx.shape[1]
        self.xg = K.x_geom(batch, 1, J, h, dh)
        if self.xg.JP > 288:
            raise NotImplementedError('cached decoding: SparseCross2DNA window too large for the single-query kernel')
        p = mod._params()
        self.nk, self.nv = p[0].detach().reshape(h, dh).contiguous(), p[1].detach().reshape(h, dh).contiguous()
        self.wth = p[2].detach().reshape(h, h).contiguous()
        self.mask_u8 = mask_u8 if mask_u8 is not None else torch.ones((batch, T), dtype=torch.uint8, device=dev)
        self.o_const = self.o_bos = self.kv = None
        if all_masked:
            kv0 = K.zeros_bf((batch * J, 2 * self.inner), dev, lo=lo)
            pk = K.xattn_pack(self.xg, kv0, self.nk, self.nv, torch.zeros((batch, J), dtype=torch.uint8, device=dev))
            self.o_const = K.xattn_decode(self.xg, K.zeros_bf((batch, self.inner), dev, lo=lo), pk, self.wth)
            self.o_bos = ops._to_bf(self.nv.float().reshape(1, self.inner).expand(batch, -1).contiguous())      # softmax over the null key alone
        else:
            W = ops.XInner.weights(mod._cache, p)
            kv = K.gemm_nt(ctx_bf, W['kv'], out_bf16=True)                   # sketch keys / values: once per sequence
            self.kv = K.BF(kv.hi.reshape(batch, T, 2 * self.inner), None if kv.lo is None else kv.lo.reshape(batch, T, 2 * self.inner))

    def attend(self, h, W, pos_dev, bos):
        """h BF [B, D] (pre-normed operand row) -> o BF [B, inner]"""
        if self.o_const is not None:
            return self.o_bos if bos else self.o_const
        q = K.gemm_nt(h, W['q'], out_bf16=True)
        B, inner, mod = self.B, self.inner, self.mod
        hd, dh = mod.heads, mod.dim_head
        if bos:
            q0 = ops._bf_val(q).reshape(B, hd, dh)
            kvf = ops._bf_val(self.kv).reshape(B, -1, 2, hd, dh)
            P0 = ops.XC2Inner._bos_scores(q0, kvf[:, :, 0], self.nk.float(), self.mask_u8.bool(), mod.scale)
            o0 = P0[..., :1] * self.nv.float()[None] + torch.einsum('bht,bthd->bhd', P0[..., 1:], kvf[:, :, 1])
            o = ops._to_bf(o0.reshape(B, inner).contiguous())
            return o if self.lo else K.BF(o.hi, None)
        i = torch.remainder(pos_dev.long() - 1, self.tpf)                    # device-side feature-map position of this row
        idx = self.win_idx.index_select(0, i)[0]
        J = idx.shape[0]
        kvw = K.BF(self.kv.hi.index_select(1, idx).reshape(B * J, 2 * inner),
                   None if self.kv.lo is None else self.kv.lo.index_select(1, idx).reshape(B * J, 2 * inner))
        m = (self.mask_u8.index_select(1, idx) * self.win_ok.index_select(0, i)).contiguous()
        pk = K.xattn_pack(self.xg, kvw, self.nk, self.nv, m)
        return K.xattn_decode(self.xg, q, pk, self.wth)


class _XmDirection:
    """One direction of a cross-modality layer (CrossModalityCrossAttention `mod`, np.py:908-1067) for row-at-a-time decoding: the
    QUERY stream calls attend(), the CONTEXT stream calls store() with the rows `mod` would see as context.  Query row r (0 = start
    token) belongs to frame (r - 1) // chunk_size and attends a learned null key + context frame f of the sequence
    [context_chunk_size - 1 zero rows, the context stream's rows]: i.e. the other stream one frame EARLIER (frame 0: zero rows + its
    start token), complete by the time any query row of frame f exists.  The start-token row outputs 0.  The Conv3d talking heads
    carry a BIAS: it adds bias[g] * (null_v[g] + sum_j v_j[g]) to the head output, a per-frame constant kept as a correction row."""

    def __init__(self, mod, batch, ctx_rows, dev, lo):
        from torch import nn
        if not (isinstance(mod.norm, nn.Identity) and isinstance(mod.context_norm, nn.Identity) and mod.has_start_token and
                mod.context_has_start_token and mod.dim_head in (32, 64) and mod.heads <= 8 and mod.context_chunk_size + 1 <= 288):
            raise NotImplementedError('cached decoding: this CrossModalityCrossAttention configuration is not on the libamdnuwa path')
        self.mod, self.B, self.lo = mod, batch, lo
        self.c, self.cc = mod.chunk_size, mod.context_chunk_size
        self.inner = mod.heads * mod.dim_head
        # context rows under mod.to_kv: cc - 1 zero rows (to_kv has no bias), then context row r at cc - 1 + r, so that context
        # frame f is rows [f * cc, (f + 1) * cc)
        self.kv = K.zeros_bf((batch, self.cc - 1 + ctx_rows, 2 * self.inner), dev, lo=lo)
        self.g = K.x_geom(batch, 1, self.cc, mod.heads, mod.dim_head)
        # n_ctx / n_q: HOST row counters (advanced by the decoder that owns the streams, never inside a captured graph);
        # pk / corr: persistent buffers, re-packed in place at every frame border (a captured step keeps reading them)
        self.n_ctx, self.n_q, self.frame = 0, 0, -1
        self.pk = K.PackedKV(self.g, dev, lo)
        self.corr = torch.zeros((batch, mod.to_out.weight.shape[0]), dtype=torch.float32, device=dev)

    def _weights(self):
        m, h = self.mod, self.mod.heads
        p = (m.null_k.detach().reshape(h, 1, -1), m.null_v.detach().reshape(h, 1, -1), m.talking_heads.weight.detach().reshape(h, h, 1, 1),
             m.to_q.weight, m.to_kv.weight, m.to_out.weight)
        return ops.XInner.weights(m._cache, p)

    def store(self, x, pos_dev):
        """x fp32 [B, D]: the context stream's row `pos_dev` (its position counter, in device memory: the write is indexed on the device
        so that the step can be replayed from a captured graph)"""
        kv = K.gemm_nt(_cast_row(x, self.lo), self._weights()['kv'], out_bf16=True)
        at = (pos_dev + (self.cc - 1)).long()
        self.kv.hi.index_copy_(1, at, kv.hi[:, None])
        if self.kv.lo is not None:
            self.kv.lo.index_copy_(1, at, kv.lo[:, None])

    def _pack(self, f):
        m, cc = self.mod, self.cc
        if self.n_ctx < f * cc + 1:
            raise RuntimeError('cached decoding: the other stream has not produced the frame this row attends')
        sl = slice(f * cc, (f + 1) * cc)
        kv = K.BF(self.kv.hi[:, sl].reshape(self.B * cc, 2 * self.inner).contiguous(),
                  self.kv.lo[:, sl].reshape(self.B * cc, 2 * self.inner).contiguous() if self.kv.lo is not None else None)
        h, dh = m.heads, m.dim_head
        K.xattn_pack(self.g, kv, m.null_k.detach().reshape(h, dh).contiguous(), m.null_v.detach().reshape(h, dh).contiguous(), None, out=self.pk)
        v = kv.hi[:, self.inner:].float()
        if kv.lo is not None:
            v = v + kv.lo[:, self.inner:].float()
        vsum = m.null_v.detach().reshape(1, h, dh) + v.reshape(self.B, cc, h, dh).sum(1)
        self.corr.copy_(F.linear((m.talking_heads.bias.detach()[None, :, None] * vsum).reshape(self.B, self.inner), m.to_out.weight.detach()))
        self.frame = f

    def needs_eager_row(self, r):
        """query row r cannot be replayed from the graph of an ordinary row: the start token (outputs 0) or the first row of a frame (packs)"""
        return r == 0 or (r - 1) // self.c != self.frame

    def attend(self, h):
        """h BF [B, D]: the operand row of query row n_q (pre-normed by the caller where the block has norms) -> fp32 [B, D]"""
        r, m = self.n_q, self.mod
        if r == 0:
            return torch.zeros((self.B, m.to_out.weight.shape[0]), dtype=torch.float32, device=h.hi.device)
        f = (r - 1) // self.c
        if f != self.frame:
            self._pack(f)
        W = self._weights()
        q = K.gemm_nt(h, W['q'], out_bf16=True)
        o = K.xattn_decode(self.g, q, self.pk, m.talking_heads.weight.detach().reshape(m.heads, m.heads).contiguous())
        return K.gemm_nt(o, W['out'], out_bf16=False) + self.corr


class DualIncrementalDecoder:
    """One pass (conditioned or text-masked) of the dual decoder -- DualModalityDecoder (np.py:1299-1487) or
    ReversibleDualModalityDecoder (np.py:1489-1655 + reversible_video_audio.py) -- over two growing sequences: step('v' | 'a', x)
    takes the decoder input row of the next position of that stream and returns the row after all layers (before the final norm)."""

    def __init__(self, dec, batch, rows_v, rows_a, context, context_mask):
        from .video_audio import DualModalityDecoder, ReversibleDualModalityDecoder
        dev, lo = context.device, K.want_lo()
        self.pos = {'v': torch.zeros(1, dtype=torch.int32, device=dev), 'a': torch.zeros(1, dtype=torch.int32, device=dev)}
        lists = {'v': [], 'a': []}
        if isinstance(dec, DualModalityDecoder):
            halves = 1
            for blocks, kind in zip(dec.layers, dec.layer_types):
                if kind == 'intra_modality':
                    for key, (attn, cross, ff) in zip('va', blocks):
                        lists[key] += [dict(mod=attn), dict(mod=cross, context=context), dict(mod=ff)]
                else:                               # both directions read the layer INPUT of the other stream (np.py:1467-1470)
                    (v_x, v_ff), (a_x, a_ff) = blocks
                    v_from_a, a_from_v = _XmDirection(v_x.fn, batch, rows_a, dev, lo), _XmDirection(a_x.fn, batch, rows_v, dev, lo)
                    lists['v'] += [dict(mod=v_x, xm=v_from_a, store_before=(a_from_v,)), dict(mod=v_ff)]
                    lists['a'] += [dict(mod=a_x, xm=a_from_v, store_before=(v_from_a,)), dict(mod=a_ff)]
        elif isinstance(dec, ReversibleDualModalityDecoder):
            halves = 2                              # y1 = x1 + f(x2), y2 = x2 + g(y1) per block; the output is the mean of the halves
            for (f, g, j, k), kind in zip(dec.layers, dec.layer_types):
                if kind == 'intra_modality_self_attn':
                    lists['v'] += [dict(mod=f, dst=0), dict(mod=g, dst=1)]
                    lists['a'] += [dict(mod=j, dst=0), dict(mod=k, dst=1)]
                elif kind == 'intra_modality_cross_attn':
                    lists['v'] += [dict(mod=f, context=context, dst=0), dict(mod=g, dst=1)]
                    lists['a'] += [dict(mod=j, context=context, dst=0), dict(mod=k, dst=1)]
                else:
                    # un-normed modules; video: y1 = x1 + f(x2, ctx = audio m2), y2 = x2 + k(y1); audio: n1 = m1 + j(m2, ctx = the
                    # UPDATED video half y2), n2 = m2 + g(n1) -- `k` / `g` crossed over as in reversible_video_audio.py:241-244
                    v_from_a, a_from_v = _XmDirection(f, batch, rows_a, dev, lo), _XmDirection(j, batch, rows_v, dev, lo)
                    lists['v'] += [dict(mod=f, xm=v_from_a, dst=0), dict(mod=k, dst=1, store_after=(a_from_v,))]
                    lists['a'] += [dict(mod=j, xm=a_from_v, dst=0, store_before=(v_from_a,)), dict(mod=g, dst=1)]
        else:
            raise NotImplementedError(type(dec).__name__)
        self.streams = {key: IncrementalDecoder(None, batch, rows, context, context_mask, self.pos[key], block_list=lists[key], halves=halves)
                        for key, rows in (('v', rows_v), ('a', rows_a))}

    def directions(self, which):
        """(_XmDirection objects this stream QUERIES through, those it STORES context rows for)"""
        blocks = self.streams[which].blocks
        return [b.xm for b in blocks if b.xm is not None], [d for b in blocks for d in tuple(b.store_before) + tuple(b.store_after)]

    def needs_eager_row(self, which):
        return any(d.needs_eager_row(d.n_q) for d in self.directions(which)[0])

    def step(self, which, x, tick=True):
        """one row of stream `which`.  tick=False: the device work only (capturable in a HIP graph: no host-side state changes); the
        caller then calls tick() once per issued or replayed row"""
        out = self.streams[which].step(x.contiguous())
        self.pos[which] += 1
        if tick:
            self.tick(which)
        return out

    def tick(self, which):
        """host bookkeeping of the row step() has just been issued (or replayed) for"""
        q, st = self.directions(which)
        for d in q:
            d.n_q += 1
        for d in st:
            d.n_ctx += 1


class DualGuidedStepper:
    """The per-token work of NUWAVideoAudio.generate: advance(which, x_row) feeds the next input row of one stream and returns the
    logits for that stream's next token; with cond_scale != 1 the final-normed conditioned output row is the input of a second,
    text-masked pass (np.py:2176-2186) and the two logits are mixed."""

    def __init__(self, model, text_embeds, text_mask, rows_v, rows_a, cond_scale, graph=True):
        self.m, self.cond_scale = model, cond_scale
        dec = model.video_audio_transformer
        B, D = text_embeds.shape[0], text_embeds.shape[-1]
        self.cond = DualIncrementalDecoder(dec, B, rows_v, rows_a, text_embeds, text_mask)
        self.uncond = DualIncrementalDecoder(dec, B, rows_v, rows_a, text_embeds, torch.zeros_like(text_mask).bool()) \
            if cond_scale != 1 else None
        # One captured HIP graph per stream for its ORDINARY rows (every row but a stream's start token and the first row of a frame,
        # where a cross-modality direction re-packs the other stream's frame on the host): static input / output buffers, positions
        # and context-row writes indexed on the device.
        self._want_graph = graph
        self.x_in = {k: torch.zeros(B, D, dtype=torch.float32, device=text_embeds.device) for k in 'va'}
        self.graphs, self.glogits = {}, {}

    def _logits(self, which, hidden):
        m, dec = self.m, self.m.video_audio_transformer
        if which == 'v':
            nrm, lin, cache = dec.video_norm.norm, m.to_video_logits, m._cache_v
        else:
            nrm, lin, cache = dec.audio_norm.norm, m.to_audio_logits, m._cache_a
        return ops.LogitsFn.apply(hidden[:, None].contiguous(), nrm.weight, nrm.bias, lin.weight, cache)[:, 0]

    def _body(self, which):
        dec = self.m.video_audio_transformer
        hidden = self.cond.step(which, self.x_in[which], tick=False)
        logits = self._logits(which, hidden)
        if self.uncond is not None:
            nrm = dec.video_norm if which == 'v' else dec.audio_norm
            uh = self.uncond.step(which, nrm(hidden[:, None])[:, 0], tick=False)
            ul = self._logits(which, uh)
            logits = ul + (logits - ul) * self.cond_scale
        return logits

    def _tick(self, which):
        self.cond.tick(which)
        if self.uncond is not None:
            self.uncond.tick(which)

    def advance(self, which, x_row):
        self.x_in[which].copy_(x_row)
        eager = not self._want_graph or self.cond.needs_eager_row(which) or (self.uncond is not None and self.uncond.needs_eager_row(which))
        if eager:
            logits = self._body(which)
            self._tick(which)
            return logits
        if which not in self.graphs:
            # warm-up on a side stream (workspaces, weight caches), positions rewound, then capture; the rows the warm-up wrote are
            # rewritten by the real step at the same positions
            decs = [self.cond] + ([self.uncond] if self.uncond is not None else [])
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._body(which)
                for d in decs:
                    d.pos[which] -= 1
            torch.cuda.current_stream().wait_stream(s)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.glogits[which] = self._body(which)
                self.graphs[which] = g
            except RuntimeError as e:
                import warnings
                warnings.warn(f'nuwa_pytorch_amd: HIP graph capture of the dual decode step failed ({e}); launching eagerly')
                self._want_graph = False
                torch.cuda.synchronize()
                logits = self._body(which)
                self._tick(which)
                return logits
        self.graphs[which].replay()
        self._tick(which)
        return self.glogits[which].clone()


class GuidedStepper:
    """The per-token work of NUWA.generate (and NUWASketch.generate, np.py:2440-2512: same decoder, sketch context): conditioned pass -> logits; if cond_scale != 1 the reference feeds the final-normed
    conditioned OUTPUT row into a second, text-masked pass (np.py:1894-1898) and mixes the two logits.  One call = one new row.
    graph=True captures the step in a HIP graph after a warm-up call (static input / output buffers, device-side position)."""

    def __init__(self, nuwa, text_embeds, text_mask, max_rows, cond_scale, graph=True):
        dev = text_embeds.device
        B, D = text_embeds.shape[0], text_embeds.shape[-1]
        self.nuwa, self.cond_scale = nuwa, cond_scale
        self.pos_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        tr = nuwa.video_transformer
        self.cond = IncrementalDecoder(tr, B, max_rows, text_embeds, text_mask, self.pos_dev)
        self.uncond = None
        if cond_scale != 1:
            self.uncond = IncrementalDecoder(tr, B, max_rows, text_embeds, torch.zeros_like(text_mask).bool(), self.pos_dev)
        self.x_in = torch.zeros(B, D, dtype=torch.float32, device=dev)
        self.logits = None
        self.graph = None
        self._want_graph = graph
        self._calls = 0

    def _body(self, bos=False):
        nuwa = self.nuwa
        hidden = self.cond.step(self.x_in, bos)
        logits = nuwa._final(hidden[:, None])[:, 0]
        if self.uncond is not None:
            cond_out = nuwa.video_transformer.norm(hidden[:, None])[:, 0].contiguous()
            uh = self.uncond.step(cond_out, bos)
            ul = nuwa._final(uh[:, None])[:, 0]
            logits = ul + (logits - ul) * self.cond_scale
        self.pos_dev += 1
        return logits

    def __call__(self, x_row):
        """x_row fp32 [B, D] = decoder input row at the current position -> logits [B, C] for the next token"""
        self.x_in.copy_(x_row)
        first = self._calls == 0
        self._calls += 1
        if first and self.cond.bos_row_differs:        # NUWASketch: the <bos> row of a SparseCross2DNA block is its own program --
            return self._body(True)                    # launched eagerly; the graph is captured at row 1 and serves every later row
        if not self._want_graph:
            return self._body()
        if self.graph is None:
            # warm-up on a side stream (weight caches, workspaces, lazy module state), then rewind the position and capture;
            # the rows the warm-up wrote are rewritten by the real step at that position
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._body()
                self.pos_dev -= 1
            torch.cuda.current_stream().wait_stream(s)
            try:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):     # capture records, it does not execute: the replay below is step `pos`
                    self.logits = self._body()
            except RuntimeError as e:                  # same kernels, launched one by one
                import warnings
                warnings.warn(f'nuwa_pytorch_amd: HIP graph capture of the decode step failed ({e}); launching eagerly')
                self.graph, self._want_graph = None, False
                torch.cuda.synchronize()
                return self._body()
        self.graph.replay()
        return self.logits
