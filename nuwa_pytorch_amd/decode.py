"""Key/value-cached token-by-token decoding for NUWA.generate (reference np.py:1841-1915).

The reference recomputes the whole prefix -- twice, for classifier-free guidance -- for every sampled token.  Every stage of
the decoder is causal (Sparse3DNA taps, np.py:420-457; ShiftVideoTokens, np.py:210-253) or row-wise (cross-attention over the
text, FeedForward, the norms), so row `pos` of the decoder depends only on rows <= pos.  IncrementalDecoder keeps, per layer,
  * the pre-norm outputs of the two token-shifted blocks (the shift reads the rows one grid row up / one token left),
  * the Sparse3DNA key / value rows,
  * the packed text keys / values of the cross-attention (computed once),
and computes ONE new row per call with the same libamdnuwa kernels the training path uses (GEMMs, LayerNorms, GEGLU, the
cross-attention core with n = 1) plus the two single-row kernels of csrc/decode.hip.  The row index lives in device memory,
so the per-token work of a whole guided step can be captured once in a HIP graph and replayed for every token."""
import torch

from . import kernels as K
from . import ops


class _Block:
    __slots__ = ('kind', 'sn', 'inner', 'fmap', 'hcache', 'kvcache', 'geom', 'pk', 'xg', 'o_const')


class IncrementalDecoder:
    """One decoder pass (conditioned or not) over a growing sequence.  transformer: nuwa_pytorch.Transformer (non-reversible)
    whose blocks are all on the fused HIP path; context [B, T, D] fp32 and context_mask [B, T] bool as in Transformer.forward."""

    def __init__(self, transformer, batch, max_rows, context, context_mask, pos_dev):
        from .nuwa_pytorch import Attention, FeedForward, Sparse3DNA
        dev = context.device
        self.B, self.rows, self.pos_dev = batch, max_rows, pos_dev
        lo = K.want_lo()
        D = context.shape[-1]
        ctx_bf = ops._ctx_to_bf(context)
        mask_u8 = context_mask.to(torch.uint8).contiguous() if context_mask is not None else None
        # the text-masked pass of classifier-free guidance: every query sees only the learned null key, so the attention
        # output is the same row for every position -- (sum_h W_th[g, h]) * null_v[g] -- and is computed once
        all_masked = context_mask is not None and not bool(context_mask.any())
        self.blocks = []
        for attn, cross, ff in transformer.layers:
            for sn, ctx_arg in ((attn, None), (cross, context), (ff, None)):
                if sn is None:
                    continue
                found = sn._inner(ctx_arg, seq_len=max_rows)
                if found is None:
                    raise NotImplementedError('IncrementalDecoder: a decoder block is not on the libamdnuwa path')
                inner, fmap = found
                blk = _Block()
                blk.sn, blk.inner, blk.fmap = sn, inner, fmap
                blk.hcache = K.zeros_bf((batch, max_rows, D), dev, lo=lo) if fmap is not None else None
                blk.kvcache = blk.geom = blk.pk = blk.xg = blk.o_const = None
                if isinstance(inner, Sparse3DNA):
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
                elif isinstance(inner, FeedForward):
                    blk.kind = 'ff'
                else:
                    raise NotImplementedError(f'IncrementalDecoder: no single-row path for {type(inner).__name__}')
                self.blocks.append(blk)

    def step(self, x):
        """x fp32 [B, D]: decoder input row `pos` of every sample -> that row after all layers (before the final norm)"""
        fast = ops._fast()
        blocks = self.blocks

        def pre(blk):
            return (blk.sn.prenorm.weight.detach(), blk.sn.prenorm.bias.detach())
        first = blocks[0]
        _, h = K.decode_ln(x, None, None, pre(first), cache=first.hcache, pos_dev=self.pos_dev, fmap=first.fmap or 0)
        for i, blk in enumerate(blocks):
            sn, inner = blk.sn, blk.inner
            p = inner._params()
            if blk.kind == 's3':
                W = ops.S3Inner.weights(inner._cache, p)
                g = blk.geom
                rel = p[5].detach().contiguous() if len(p) > 5 else None
                qkv = K.gemm_nt(h, W['qkv'], out_bf16=True)
                o = K.s3_decode(g, qkv, blk.kvcache, self.pos_dev, p[2].detach().reshape(g.heads, g.heads).contiguous(), rel)
                y = K.gemm_nt(o, W['out'], bias=p[4].detach(), out_bf16=fast)
            elif blk.kind == 'x':
                W = ops.XInner.weights(inner._cache, p)
                g = blk.xg
                wth = p[2].detach().reshape(g.heads, g.heads).contiguous()
                if blk.o_const is not None:
                    o = blk.o_const
                else:
                    q = K.gemm_nt(h, W['q'], out_bf16=True)
                    o = K.xattn_decode(g, q, blk.pk, wth)
                y = K.gemm_nt(o, W['out'], out_bf16=fast)
            else:
                W = ops.FFInner.weights(inner._cache, p)
                u = K.gemm_nt(h, W['w1'], out_bf16=True)
                gg = K.geglu_fwd(u, W['FP'], interleaved=True)
                y = K.gemm_nt(gg, W['w2'], out_bf16=fast)
            # post-norm + residual, the next block's pre-norm and its token shift (cache write + gather): one launch
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            x, h = K.decode_ln(y, x, (sn.postnorm.weight.detach(), sn.postnorm.bias.detach()), pre(nxt) if nxt else None,
                               cache=nxt.hcache if nxt else None, pos_dev=self.pos_dev, fmap=(nxt.fmap or 0) if nxt else 0)
        return x


class GuidedStepper:
    """The per-token work of NUWA.generate: conditioned pass -> logits; if cond_scale != 1 the reference feeds the final-normed
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

    def _body(self):
        nuwa = self.nuwa
        hidden = self.cond.step(self.x_in)
        logits = nuwa._final(hidden[:, None])[:, 0]
        if self.uncond is not None:
            cond_out = nuwa.video_transformer.norm(hidden[:, None])[:, 0].contiguous()
            uh = self.uncond.step(cond_out)
            ul = nuwa._final(uh[:, None])[:, 0]
            logits = ul + (logits - ul) * self.cond_scale
        self.pos_dev += 1
        return logits

    def __call__(self, x_row):
        """x_row fp32 [B, D] = decoder input row at the current position -> logits [B, C] for the next token"""
        self.x_in.copy_(x_row)
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
