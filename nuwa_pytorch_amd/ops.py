"""autograd.Function wrappers that sequence the libamdnuwa kernels for the decoder sub-blocks.

Each decoder sub-block of the reference (Transformer.forward np.py:1174-1180)

        x = SandwichNorm(fn)(x, ...) + x          fn in {Shift(Sparse3DNA), Attention(context), Shift(FeedForward)}

is ONE autograd node here (`SandwichBlockFn`): pre-LN -> GEMM(s) [token shift folded into the GEMM
loader] -> attention core / GEGLU -> GEMM -> post-LN + residual, with a hand-written backward that
calls the backward kernels.  The residual stream, LayerNorm statistics, softmax and all accumulators
are fp32; GEMM / attention operands are bf16 (plus a bf16 residual in 'bf16x3' parity mode).

The same inner stages also back the standalone modules (Sparse3DNA, Attention, FeedForward,
LayerNorm wrappers) through `InnerFn`, so `Sparse3DNA(...)(x)` alone works as in the reference.
"""
import functools
import os
import warnings
import weakref

import torch
from torch.autograd import Function

from . import kernels as K
from .kernels import BF


def _ru(v, m):
    return (v + m - 1) // m * m


def _in_phase(name):
    """run an autograd node's forward / backward body inside K.phase(name) (see kernels.phase: the mixed precision mode allocates
    and consumes hi-only operands in backward code)"""
    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            with K.phase(name):
                return fn(*a, **k)
        return wrapped
    return deco


class WeightCache:
    """bf16 (hi[/lo]) operand copies of fp32 master weights; rebuilt when a parameter changes
    (optimizer step bumps `_version`) or the precision mode changes."""

    EPOCH = 0          # bumped by optimisers that update parameters behind autograd's back (FusedAdamW)

    def __init__(self):
        self._store = {}

    def get(self, key, params, builder):
        ver = tuple((p.data_ptr(), p._version) for p in params) + (K.get_precision(), WeightCache.EPOCH, K._BWD_F16)
        ent = self._store.get(key)
        if ent is None or ent[0] != ver:
            with torch.no_grad():
                ent = (ver, builder())
            self._store[key] = ent
        return ent[1]

    def clear(self):
        self._store.clear()


def _cast(w):
    out = K.empty_bf(tuple(w.shape), w.device)
    K.cast_pad(w.detach(), out)
    return out


def _cast_t(w):
    out = K.empty_bf((w.shape[1], w.shape[0]), w.device)
    K.transpose_cast(w.detach(), out)
    return out


# fp16 range guard of the 'bf16x3-fwd' forward (its FeedForward / 3DNA q-k-v products run single fp16 MFMAs on fp16 copies of the
# weights).  fp16 ends at 65504 -- a larger weight would become inf -- and loses significand bits below 6.1e-5, so a weight
# tensor takes the fp16 form only while its largest magnitude lies in [F16_WMIN, F16_WMAX]; otherwise the block runs the bf16 hi + lo
# (3-MFMA) products, which have fp32's range.  One device -> host transfer per call of f16_ranges_prefetch() covers every weight
# whose (storage, version) has not been judged yet: Transformer.forward_layers calls it once per step for the whole stack.
# (Activations are covered in the kernels: every fp16 activation store saturates at +-65504, csrc/common.h pack2_f16_sat.)
F16_WMAX = 6.0e4
F16_WMIN = 2.0 ** -10
_F16_RANGE = {}


_F16_ALSO = []


def f16_prefetch_also(*ws):
    """weights to be judged together with the next f16_ranges_prefetch() call (no transfer of their own): NUWA.forward registers
    to_logits.weight here before the decoder stack runs"""
    _F16_ALSO.extend(ws)


def f16_ranges_prefetch(ws):
    todo, seen = [], set()
    ws = list(ws) + _F16_ALSO
    del _F16_ALSO[:]
    for w in ws:
        k = (w.data_ptr(), w._version)
        if not _f16_known(w) and k not in seen and w.is_cuda:
            seen.add(k)
            todo.append(w)
    if not todo:
        return
    if len(_F16_RANGE) > 8192:
        _F16_RANGE.clear()
    with torch.no_grad():
        am = torch.stack(torch._foreach_norm([w.detach() for w in todo], float('inf'))).cpu().tolist()     # ONE synchronisation
    for w, a in zip(todo, am):
        _F16_RANGE[(w.data_ptr(), w._version)] = (weakref.ref(w), bool(F16_WMIN <= a <= F16_WMAX))       # (NaN compares false)
    # the stream is drained at this point anyway (once per optimiser step): look at the fp16 saturation monitor of the step before
    nsat = K.f16_sat_count(reset=True)
    if nsat:
        warnings.warn(f'nuwa_pytorch_amd: {nsat} thread(s) clamped a value beyond +-65504 in a counted fp16 store of the last step (LayerNorm copies, the cross-attention '
                      "output copy or fp16 gradients of the 'bf16x3-fwd' mode; the q / k / v and gate copies clamp without counting): the step ran on saturated values; set_precision('bf16x3') has fp32's range",
                      RuntimeWarning, stacklevel=3)


def _f16_known(w):
    """the verdict recorded for THIS tensor object at this version, or None: a verdict is tied to the tensor by a weak reference, so a new
    weight allocated at a freed weight's address (same data_ptr, same version counter) does not inherit a stale one"""
    ent = _F16_RANGE.get((w.data_ptr(), w._version))
    return ent[1] if ent is not None and ent[0]() is w else None


def f16_weights_ok(*ws):
    f16_ranges_prefetch(ws)
    return all(_f16_known(w) is True for w in ws)


def _f16_to_pair(h):
    """an activation that arrived as a bf16 + fp16 copy pair, for a block that has to run hi + lo products after all (its weights
    left the fp16 range): hi + lo with lo = bf16(fp16 value - hi) carries the fp16 value exactly (11 significand bits)"""
    if h.lo is None and h.f16 is not None:
        return BF(h.hi, (h.f16.float() - h.hi.float()).to(torch.bfloat16))
    return h


# =================================================================================================
# inner stages.  fwd(h: BF [R, D], ...) -> (y fp32 [R, D], saved) ; bwd(saved, dy: BF) -> (dh fp32, grads)
# =================================================================================================

class S3Inner:
    """to_q/to_kv -> Sparse3DNA core -> to_out (np.py:481-613).  params: to_q.w, to_kv.w, talking_heads.w,
    to_out.w, to_out.b"""
    nparams = 5

    @staticmethod
    def weights(cache, p):
        wq, wkv, wth, wo, bo = p[:5]

        def build():
            inner, D = wq.shape
            qkv = K.empty_bf((3 * inner, D), wq.device)
            K.cast_pad(wq.detach(), qkv, row0=0)
            K.cast_pad(wkv.detach(), qkv, row0=inner)
            qkvT = K.empty_bf((D, 3 * inner), wq.device)
            K.transpose_cast(wq.detach(), qkvT, col0=0)
            K.transpose_cast(wkv.detach(), qkvT, col0=inner)
            out = dict(qkv=qkv, qkvT=qkvT, out=_cast(wo), outT=_cast_t(wo))
            if K.mixed() and f16_weights_ok(wq, wkv):   # fp16 copy for the fp16-operand q / k / v projection of 'bf16x3-fwd'
                out['qkv_16'] = torch.cat((wq.detach(), wkv.detach()), 0).to(torch.float16).contiguous()
            if K.mixed() and f16_weights_ok(wo):        # fp16 hi + lo pair for the two-MFMA to_out product
                out['out_16'] = K.f16_pair(wo)
            if K.bwd_f16('s') and 'qkv_16' in out and 'out_16' in out:    # transposes for the dgrad products of the fp16-gradient backward
                out['qkvT_16'] = out['qkv_16'].t().contiguous()
                out['outT_16'] = wo.detach().t().contiguous().to(torch.float16)
            return out
        return cache.get('s3', (wq, wkv, wo), build)

    @staticmethod
    def bwd16_ok(R, D, inner, g, ws=None, wo=None, rel=False):
        """the fp16-gradient backward applies (class 's'): the fp16 projection + core + two-MFMA to_out of the forward apply, the band kernels take
        the geometry without a relative-position bias, and the five backward products fit their fp16 kernels"""
        return K.bwd_f16('s') and not rel and S3Inner.f16_proj_ok(R, D, inner, g, ws) and K.proj_f16x2('o') and \
            K.gemm_nt_f16x2_ok(R, D, inner, out_bf16=False) and (wo is None or f16_weights_ok(wo)) and K.s3_bwd16_supported(g) and \
            K.gemm_nt_f16ops_ok(R, 3 * inner, D, out_bf16=False, out_f16=True) and K.gemm_nt_f16ops_ok(R, inner, D, out_bf16=False, out_f16=True) and \
            K.gemm_nt_f16ops_ok(R, D, 3 * inner, out_bf16=False, out_f16=True) and K.gemm_tn16_ok(R, D, inner) and K.gemm_tn16_ok(R, 3 * inner, D)

    @staticmethod
    def f16_proj_ok(R, D, inner, g, ws=None):
        """'bf16x3-fwd': q / k / v projection on single fp16 MFMAs (bf16 + fp16 copies out) feeding the fp16 core; ws = (to_q.weight,
        to_kv.weight) adds the fp16 range guard of the weights"""
        return K.qkv_f16() and K.s3_f16_supported(g) and K.gemm_nt_f16ops_ok(R, 3 * inner, D, out_bf16=True) and \
            (ws is None or f16_weights_ok(*ws))

    @staticmethod
    def fwd(h, p, meta):
        W = S3Inner.weights(meta['cache'], p)
        wth, bo = p[2], p[4]
        g = meta['geom']
        rel = p[5].detach().contiguous() if len(p) > 5 else None          # [J, heads] relative-position bias (optional)
        # 'bf16x3-fwd': q / k / v leave the 3-MFMA projection as a bf16 copy (backward) + an fp16 copy, and the core runs single fp16 MFMAs
        R, D, _ = K.bf_rows_cols(h)
        if meta.get('bwd16') and 'qkvT_16' in W and rel is None:
            # fp16-gradient backward: ONE (fp16) copy of h, of q / k / v and of the core's output
            qkv = K.BF(None, None, K.gemm_nt_f16ops(h.f16, W['qkv_16'], out_f16=True))
            o = K.sparse3dna_fwd(g, qkv, wth.detach().reshape(g.heads, g.heads).contiguous(), o_f16='only')
            y = K.gemm_nt_f16x2(o.f16, W['out_16'], bias=bo.detach())
            return y, (K.BF(None, None, h.f16), qkv, o)
        assert h.hi is not None, 'a bf16 backward needs the bf16 copy of the LayerNorm output'
        if meta.get('shift') is None and 'qkv_16' in W and S3Inner.f16_proj_ok(R, D, g.heads * g.dim_head, g):
            h16 = h.f16 if h.f16 is not None else K.hilo_to_f16(h)
            qkv = K.gemm_nt_f16ops(h16, W['qkv_16'], out_bf16=True, copy_f16=True)
        else:
            h = _f16_to_pair(h)
            f16 = K.cores_f16() and h.lo is not None and K.s3_f16_supported(g)
            qkv = K.gemm_nt(h, W['qkv'], out_bf16=True, shift=meta.get('shift'), out_f16=f16)
        # two-MFMA to_out: the fp16 core hands its output over as a bf16 copy (backward) + an fp16 copy (this product's A operand)
        o16 = K.proj_f16x2('o') and qkv.f16 is not None and 'out_16' in W and K.gemm_nt_f16x2_ok(R, p[3].shape[0], p[3].shape[1], out_bf16=False)
        o = K.sparse3dna_fwd(g, qkv, wth.detach().reshape(g.heads, g.heads).contiguous(), rel_bias=rel, o_f16=o16)
        y = K.gemm_nt_f16x2(o.f16, W['out_16'], bias=bo.detach()) if o16 else K.gemm_nt(o, W['out'], bias=bo.detach(), out_bf16=_fast())
        return y, _sv(h, qkv, o)

    @staticmethod
    def bwd(saved, dy, p, meta, need_dbias=True, dy_f32=None):
        h, qkv, o = saved
        W = S3Inner.weights(meta['cache'], p)
        wq, wkv, wth, wo, bo = p[:5]
        rel = p[5].detach().contiguous() if len(p) > 5 else None
        g = meta['geom']
        inner = g.heads * g.dim_head
        if isinstance(dy, K.G16):
            # fp16-gradient backward: dy = fp16(S dy); every product on the fp16 MFMA against the fp16 copies the forward left
            s2 = dy.s2
            d_o = K.gemm_nt_f16ops(dy.t, W['outT_16'], out_f16=True)
            dwo = torch.empty_like(wo)
            K.gemm_tn16(dy.t, o.f16, dwo, s2)
            dqkv, dwth = K.sparse3dna_bwd16(g, qkv.f16, wth.detach().reshape(g.heads, g.heads).contiguous(), d_o, s2)
            dh = K.G16(K.gemm_nt_f16ops(dqkv, W['qkvT_16'], out_f16=True), s2)
            dwqkv = torch.empty((3 * inner, wq.shape[1]), dtype=torch.float32, device=wq.device)
            K.gemm_tn16(dqkv, h.f16, dwqkv, s2)
            return dh, None, [dwqkv[:inner], dwqkv[inner:], dwth.reshape(wth.shape), dwo, None]
        d_o = K.gemm_nt(dy, W['outT'], out_bf16=True)
        dwo = torch.empty_like(wo)
        K.gemm_tn(dy, o, dwo)
        dqkv, dwth, drel = K.sparse3dna_bwd(g, qkv, wth.detach().reshape(g.heads, g.heads).contiguous(), d_o, rel_bias=rel)
        dh = K.gemm_nt(dqkv, W['qkvT'], out_bf16=_fast_bwd())
        sh = meta.get('shift')
        dwqkv = torch.empty((3 * inner, wq.shape[1]), dtype=torch.float32, device=wq.device)   # one wgrad GEMM for [to_q; to_kv]
        K.gemm_tn(dqkv, h, dwqkv, shift=sh)
        dwq, dwkv = dwqkv[:inner], dwqkv[inner:]
        dbo = K.colsum(dy_f32) if (need_dbias and dy_f32 is not None) else None
        grads = [dwq, dwkv, dwth.reshape(wth.shape), dwo, dbo]
        if rel is not None:
            grads.append(drel)
        return dh, None, grads


def _rotary_bf(t, freqs, B, n, heads_total, inverse=False):
    """apply_rotary_pos_emb (np.py:144-153) to a bf16 [B*n, heads_total*dh] activation (every dh-wide head chunk: q, k AND v
    heads alike -- quirk Q11); inverse=True applies the transpose (the backward of the rotation).  Text-encoder sized tensors."""
    x = t.hi.float() if t.lo is None else t.hi.float() + t.lo.float()
    R, Cc = x.shape
    x = x.view(B, n, heads_total, Cc // heads_total)
    rd = freqs.shape[-1]
    c, sn = freqs.cos()[None, :, None, :], freqs.sin()[None, :, None, :]
    a, rest = x[..., :rd], x[..., rd:]
    rh = torch.cat((-a[..., rd // 2:], a[..., :rd // 2]), dim=-1)
    a = a * c - rh * sn if inverse else a * c + rh * sn
    out = K.empty_bf((R, Cc), x.device, lo=t.lo is not None)
    K.cast_pad(torch.cat((a, rest), dim=-1).reshape(R, Cc).contiguous(), out)
    return out


class XInner:
    """to_q(x), to_kv(context) -> cross-attention core -> to_out (np.py:315-379, context given).
    params: null_k, null_v, talking_heads.w, to_q.w, to_kv.w, to_out.w"""
    nparams = 6

    @staticmethod
    def weights(cache, p):
        nk, nv, wth, wq, wkv, wo = p

        def build():
            out = dict(q=_cast(wq), qT=_cast_t(wq), kv=_cast(wkv), kvT=_cast_t(wkv), out=_cast(wo), outT=_cast_t(wo))
            if K.mixed() and f16_weights_ok(wq, wo):    # fp16 hi + lo pairs for the two-MFMA q and to_out products of 'bf16x3-fwd'
                out['q_16'], out['out_16'] = K.f16_pair(wq), K.f16_pair(wo)
                if K.bwd_f16('x'):                      # ... and fp16 transposes for the dgrad products of the fp16-gradient backward
                    out['qT_16'] = wq.detach().t().contiguous().to(torch.float16)
                    out['outT_16'] = wo.detach().t().contiguous().to(torch.float16)
            return out
        return cache.get('x', (wq, wkv, wo), build)

    @staticmethod
    def bwd16_ok(R, D, inner, g, meta, ws=None):
        """the fp16-gradient backward applies (class 'x'): the two-MFMA q projection, the fp16 xattn6 cores and the two-MFMA to_out of the forward
        apply, and the backward products fit their fp16 kernels (ws = to_q.weight, to_out.weight)"""
        return K.bwd_f16('x') and XInner.f16x2_ok(R, D, inner, g, meta, ws) and K.proj_f16x2('o') and K.xattn6_on() and K.xattn6_supported(g) and \
            K.gemm_nt_f16x2_ok(R, D, inner, out_bf16=False) and K.xattn_bwd16_ok(g) and not K.xattn2_bwd_rc_ok(g) and \
            K.gemm_nt_f16ops_ok(R, inner, D, out_bf16=False, out_f16=True) and K.gemm_nt_f16ops_ok(R, D, inner, out_bf16=False, out_f16=True) and \
            K.gemm_tn16_ok(R, D, inner) and K.gemm_tn16_ok(R, inner, D)

    @staticmethod
    def f16x2_ok(R, D, inner, g, meta, ws=None):
        """'bf16x3-fwd' with the two-MFMA switch on: the q projection takes the LayerNorm output as ONE fp16 value (so the LayerNorm store
        should write a bf16 + fp16 copy pair) -- when the fp16 core follows, the product fits the ring and (ws = to_q.weight, to_out.weight)
        the weights sit inside the fp16 range"""
        return K.proj_f16x2('q') and K.cores_f16() and not meta.get('self_kv') and meta.get('rotary') is None and K.xattn2_supported(g) and \
            K.gemm_nt_f16x2_ok(R, inner, D, out_bf16=True) and (ws is None or f16_weights_ok(*ws))

    @staticmethod
    def fwd(h, p, meta):
        W = XInner.weights(meta['cache'], p)
        nk, nv, wth = p[0], p[1], p[2]
        g = meta['xgeom']
        ctx = h if meta.get('self_kv') else meta['ctx_bf']          # self-attention (text encoder): keys / values from the same rows
        rot = meta.get('rotary')
        R, D, _ = K.bf_rows_cols(h)
        inner = g.heads * g.dim_head
        if meta.get('bwd16') and 'qT_16' in W and ctx.lo is not None and nk.dtype == torch.float32:
            # fp16-gradient backward: ONE (fp16) copy of h, of q and of the core's output; the backward's images come from the fp16 copy of k / v
            q16 = K.gemm_nt_f16x2(h.f16, W['q_16'], out_f16=True)
            kv = K.gemm_nt(ctx, W['kv'], out_bf16=True, out_f16=True)
            nk2, nv2 = nk.detach().reshape(g.heads, g.dim_head).contiguous(), nv.detach().reshape(g.heads, g.dim_head).contiguous()
            wth2 = wth.detach().reshape(g.heads, g.heads).contiguous()
            pk = K.xattn6_pack_bwd(g, kv.f16, nk2, nv2, meta['mask_u8'])
            o, stats = K.xattn6_fwd(g, q16, K.xattn6_pack(g, kv.f16, meta['mask_u8']), nk2, nv2, wth2, o_f16='only')
            y = K.gemm_nt_f16x2(o.f16, W['out_16'])
            return y, (K.BF(None, None, h.f16), K.hi_only(ctx), K.BF(None, None, q16), pk, stats, None, o)
        assert h.hi is not None, 'a bf16 backward needs the bf16 copy of the LayerNorm output'
        x2 = 'q_16' in W and ctx.lo is not None and XInner.f16x2_ok(R, D, inner, g, meta)
        if not x2:
            h = _f16_to_pair(h)
        f16 = K.cores_f16() and (x2 or h.lo is not None) and ctx.lo is not None and rot is None and K.xattn2_supported(g)
        if x2:      # two MFMAs: fp16 LayerNorm output x the weight as an fp16 hi + lo pair; q leaves as a bf16 copy + an fp16 copy
            q = K.gemm_nt_f16x2(h.f16 if h.f16 is not None else K.hilo_to_f16(h), W['q_16'], out_bf16=True, copy_f16=True)
        else:
            q = K.gemm_nt(h, W['q'], out_bf16=True, out_f16=f16)
        kv = K.gemm_nt(ctx, W['kv'], out_bf16=True, out_f16=f16)          # (context rows: B * T of them, a 3-MFMA product either way)
        if rot is not None:
            q = _rotary_bf(q, rot, g.B, g.n, g.heads)
            kv = _rotary_bf(kv, rot, g.B, g.T, 2 * g.heads)
        nk2, nv2 = nk.detach().reshape(g.heads, g.dim_head).contiguous(), nv.detach().reshape(g.heads, g.dim_head).contiguous()
        # third-design forward core (xattn6: images in LDS order, null key as a rank-one term) wherever the second design ran; the bf16
        # recomputing backward keeps its two [key][d] images
        x6 = K.xattn6_on() and K.xattn6_supported(g) and (f16 or K.xattn2_supported(g, q)) and nk2.dtype == torch.float32
        x6b = x6 and K.xattn6_bwd_ok(g) and not K.xattn2_bwd_rc_ok(g)
        if x6b:       # the backward's own images (bf16 [key][d] tiles in LDS order)
            pk = K.xattn6_pack_bwd(g, kv.hi, nk2, nv2, meta['mask_u8'])
        else:
            pk = K.xattn_pack(g, kv, nk2, nv2, meta['mask_u8'], lean=('bwd' if x6 and not K.xattn2_bwd_rc_ok(g) else (f16 and not K.xattn2_bwd_rc_ok(g))))
        wth2 = wth.detach().reshape(g.heads, g.heads).contiguous()
        o16 = False
        if f16:                               # 'bf16x3-fwd': the forward core on single fp16 MFMAs, hi + lo output, statistics for the bf16 backward
            o16 = K.proj_f16x2('o') and 'out_16' in W and K.gemm_nt_f16x2_ok(R, p[5].shape[0], p[5].shape[1], out_bf16=False)
            if x6:
                o, stats = K.xattn6_fwd(g, q.f16, K.xattn6_pack(g, kv.f16, meta['mask_u8']), nk2, nv2, wth2, o_f16=o16)
            else:
                o, stats = K.xattn2_fwd_f16(g, q, pk, wth2, o_f16=o16)
                pk.drop_lo()
            P, Pm = stats, None
        elif K.xattn2_supported(g, q):        # fast mode: statistics only, the backward recomputes the probabilities
            if x6:
                o, stats = K.xattn6_fwd(g, q.hi, K.xattn6_pack(g, kv.hi, meta['mask_u8']), nk2, nv2, wth2, lo=False)
            else:
                o, stats = K.xattn2_fwd(g, q, pk, wth2)
            P, Pm = stats, None
        elif K.mixed() and K.xattn2_supported(g):
            # 3-MFMA forward that leaves only the softmax statistics; the bf16 backward (xattn2_bwd) recomputes from the hi parts
            o, stats = K.xattn_fwd(g, q, pk, wth2, want_stats=True)
            P, Pm = stats, None
            pk.drop_lo()
        else:
            o, P, Pm = K.xattn_fwd(g, q, pk, wth2, save=meta.get('save', True))
        y = K.gemm_nt_f16x2(o.f16, W['out_16']) if o16 else K.gemm_nt(o, W['out'], out_bf16=_fast())
        return y, _sv(h, ctx, q, pk, P, Pm, o)

    @staticmethod
    def bwd(saved, dy, p, meta, need_dbias=False, dy_f32=None):
        h, ctx, q, pk, P, Pm, o = saved
        W = XInner.weights(meta['cache'], p)
        nk, nv, wth, wq, wkv, wo = p
        g = meta['xgeom']
        wth2 = wth.detach().reshape(g.heads, g.heads).contiguous()
        if isinstance(dy, K.G16):
            # fp16-gradient backward: dy = fp16(S dy); every large product on the fp16 MFMA against the fp16 copies the forward left
            s2 = dy.s2
            d_o = K.gemm_nt_f16ops(dy.t, W['outT_16'], out_f16=True)
            dwo = torch.empty_like(wo)
            K.gemm_tn16(dy.t, o.f16, dwo, s2)
            dq, dS, Pm, dwth = K.xattn6_bwd16(g, q.f16, d_o, pk, wth2, P, s2)
            dKp, dVp = K.xattn_kv_grads16(g, dS, Pm, q.f16, d_o, s2)
            dkv, dnk, dnv = K.xattn_unpack(g, dKp, dVp, lo=False, permuted=True, null_last=True)       # (context-sized from here on: bf16 as before)
            dh = K.G16(K.gemm_nt_f16ops(dq, W['qT_16'], out_f16=True), s2)
            dwq, dwkv = torch.empty_like(wq), torch.empty_like(wkv)
            K.gemm_tn16(dq, h.f16, dwq, s2)
            K.gemm_tn(dkv, ctx, dwkv)
            dctx = K.gemm_nt(dkv, W['kvT'])
            return dh, dctx, [dnk.reshape(nk.shape), dnv.reshape(nv.shape), dwth.reshape(wth.shape), dwq, dwkv, dwo]
        d_o = K.gemm_nt(dy, W['outT'], out_bf16=True)
        dwo = torch.empty_like(wo)
        K.gemm_tn(dy, o, dwo)
        permuted = False
        if Pm is None and K.xattn2_bwd_rc_ok(g):
            dq, dKp, dVp, dwth = K.xattn2_bwd_rc(g, q, d_o, pk, wth2, P)         # no dS / Pm arrays: the key side recomputes them
        else:
            if Pm is None:
                if isinstance(pk, K.PackedKV6B):
                    dq, dS, Pm, dwth = K.xattn6_bwd(g, q, d_o, pk, wth2, P)
                else:
                    dq, dS, Pm, dwth = K.xattn2_bwd(g, q, d_o, pk, wth2, P)      # dS / Pm columns in the kernel's chunk-permuted key order
                permuted = True
            else:
                dq, dS, dwth = K.xattn_bwd(g, d_o, pk, wth2, P)
            dKp, dVp = K.xattn_kv_grads(g, dS, Pm, q, d_o)
        dkv, dnk, dnv = K.xattn_unpack(g, dKp, dVp, lo=dy.lo is not None, permuted=permuted, null_last=isinstance(pk, K.PackedKV6B))
        rot = meta.get('rotary')
        if rot is not None:
            dq = _rotary_bf(dq, rot, g.B, g.n, g.heads, inverse=True)
            dkv = _rotary_bf(dkv, rot, g.B, g.T, 2 * g.heads, inverse=True)
        dh = K.gemm_nt(dq, W['qT'], out_bf16=_fast_bwd())
        dwq, dwkv = torch.empty_like(wq), torch.empty_like(wkv)
        K.gemm_tn(dq, h, dwq)
        K.gemm_tn(dkv, ctx, dwkv)
        dctx = K.gemm_nt(dkv, W['kvT'])
        if meta.get('self_kv'):                  # the key/value rows ARE the query rows: one gradient for h
            dh = _as_f32(dh) + dctx
            dctx = None
        return dh, dctx, [dnk.reshape(nk.shape), dnv.reshape(nv.shape), dwth.reshape(wth.shape), dwq, dwkv, dwo]


def _ff_keep_mask(R, C, p, device):
    """keep mask of nn.Dropout(p) over the GEGLU output [R, C] (torch's RNG stream; tests replace this function by a fixed mask)"""
    return torch.rand((R, C), device=device) >= p


def _ff_drop(t_f32, keep, p):
    """nn.Dropout in training: kept entries times 1 / (1 - p) (one fp32 multiply, as torch does), the others exactly 0"""
    return torch.where(keep, t_f32 * (1.0 / (1.0 - p)), torch.zeros((), dtype=t_f32.dtype, device=t_f32.device))


class FFInner:
    """Linear -> GEGLU -> Linear (np.py:255-286).  params: net.0.w (2*FFI, D), net.3.w (D, FFI).
    The inner width is zero-padded to FP = roundup(FFI, 32) inside the bf16 copies."""
    nparams = 2

    @staticmethod
    def weights(cache, p):
        w1, w2 = p

        def build():
            D, FFI = w2.shape
            FP = _ru(FFI, 32)
            dev = w1.device
            # FF1 rows in the interleaved-by-8 order (8 value rows, their 8 gate rows, ...): a lane of the GEMM epilogue owns 16
            # contiguous output columns, so it holds values AND gates and can apply the GEGLU gate itself
            w1pad = torch.zeros((2 * FP, D), dtype=torch.float32, device=dev)
            w1pad[:FFI] = w1.detach()[:FFI]
            w1pad[FP:FP + FFI] = w1.detach()[FFI:]
            w1il = K.geglu_interleave(w1pad, FP, dim=0).contiguous()
            w1p = K.zeros_bf((2 * FP, D), dev)
            K.cast_pad(w1il, w1p, row0=0)
            w1T = K.zeros_bf((D, 2 * FP), dev)
            K.transpose_cast(w1il, w1T, col0=0)
            w2p = K.zeros_bf((D, FP), dev)
            K.cast_pad(w2.detach(), w2p, Cp=FP)
            w2T = K.zeros_bf((FP, D), dev)
            K.transpose_cast(w2.detach(), w2T)
            out = dict(w1=w1p, w1T=w1T, w2=w2p, w2T=w2T, FP=FP, FFI=FFI)
            if K.mixed() and f16_weights_ok(w1, w2):    # fp16 copies for the fp16-operand forward GEMMs of 'bf16x3-fwd'
                out['w1_16'] = w1il.to(torch.float16).contiguous()
                w2pad = torch.zeros((D, FP), dtype=torch.float32, device=dev)
                w2pad[:, :FFI] = w2.detach()
                out['w2_16'] = w2pad.to(torch.float16)
                if K.bwd_f16('f'):                      # ... and their transposes for the dgrad products of the fp16-gradient backward
                    out['w1T_16'] = out['w1_16'].t().contiguous()
                    out['w2T_16'] = out['w2_16'].t().contiguous()
            return out
        return cache.get('ff', (w1, w2), build)

    @staticmethod
    def f16_ok(R, D, FP, ws=None):
        """the fp16-operand forward applies when 'bf16x3-fwd' wants it, both products run on the 256x256 ring and (ws = the two
        weights) the weights sit inside the fp16 range"""
        return K.ff_f16() and K.gemm_nt_f16ops_ok(R, 2 * FP, D, out_bf16=True, gate=True) and K.gemm_nt_f16ops_ok(R, D, FP, out_bf16=False) and \
            (ws is None or f16_weights_ok(*ws))

    @staticmethod
    def bwd16_ok(R, D, FP, FFI, ws=None):
        """the fp16-gradient backward applies: switched on, the fp16 forward applies, and the four backward products fit their fp16 kernels"""
        return K.bwd_f16('f') and FFInner.f16_ok(R, D, FP, ws) and \
            K.gemm_nt_f16ops_ok(R, FP, D, out_bf16=True, geglu_bwd=True) and K.gemm_nt_f16ops_ok(R, D, 2 * FP, out_bf16=False, out_f16=True) and \
            K.gemm_tn16_ok(R, D, FFI, lda=D, ldb=FP) and K.gemm_tn16_ok(R, 2 * FP, D)

    @staticmethod
    def fwd(h, p, meta):
        W = FFInner.weights(meta['cache'], p)
        R, D, _ = K.bf_rows_cols(h)
        drop_p = float(meta.get('drop_p') or 0.0)
        assert not (drop_p and meta.get('bwd16')), 'FeedForward dropout runs on the bf16 backward (the fp16 GEGLU-backward epilogue has no mask input)'
        if meta.get('shift') is None and 'w1_16' in W and FFInner.f16_ok(R, D, W['FP']):
            # 'bf16x3-fwd': both FeedForward products on single fp16 MFMAs (h arrives with an fp16 copy from the LayerNorm store);
            # u and the gate output also leave as bf16 copies for the bf16 backward
            h16 = h.f16 if h.f16 is not None else K.hilo_to_f16(h)
            if meta.get('bwd16') and 'w1T_16' in W:
                # fp16-gradient backward: ONE copy of h and of the gate output (fp16); u stays bf16 (read element-wise by the gate's backward)
                u, gg16, _ = K.gemm_nt_f16ops(h16, W['w1_16'], out_bf16=True, gate=True, gate_bf16=False)
                y = K.gemm_nt_f16ops(gg16, W['w2_16'])
                return y, (K.BF(None, None, h16), K.BF(u, None), K.BF(None, None, gg16))
            assert h.hi is not None, 'a bf16 backward needs the bf16 copy of the LayerNorm output'
            u, gg16, ggb = K.gemm_nt_f16ops(h16, W['w1_16'], out_bf16=True, gate=True)
            if drop_p:          # ff_dropout > 0 in training (np.py:276): nn.Dropout on the GEGLU output, between the two products
                keep = _ff_keep_mask(R, W['FP'], drop_p, gg16.device)
                gf = _ff_drop(gg16.float(), keep, drop_p)
                gg16, ggb = gf.to(torch.float16), gf.to(torch.bfloat16)
                return K.gemm_nt_f16ops(gg16, W['w2_16']), (K.BF(h.hi, None), K.BF(u, None), K.BF(ggb, None), keep)
            y = K.gemm_nt_f16ops(gg16, W['w2_16'])
            return y, (K.BF(h.hi, None), K.BF(u, None), K.BF(ggb, None))
        h = _f16_to_pair(h)
        gg = K.empty_bf((h.hi.shape[0], W['FP']), h.hi.device)
        u = K.gemm_nt(h, W['w1'], out_bf16=True, shift=meta.get('shift'), geglu_out=gg)   # u (interleaved layout) and a * gelu(gate)
        if drop_p:
            keep = _ff_keep_mask(R, W['FP'], drop_p, h.hi.device)
            ggd = K.empty_bf((h.hi.shape[0], W['FP']), h.hi.device)
            K.cast_pad(_ff_drop(_bf_val(gg), keep, drop_p).contiguous(), ggd)
            return K.gemm_nt(ggd, W['w2'], out_bf16=_fast()), (*_sv(h, u, ggd), keep)
        y = K.gemm_nt(gg, W['w2'], out_bf16=_fast())
        return y, _sv(h, u, gg)

    @staticmethod
    def bwd(saved, dy, p, meta, need_dbias=False, dy_f32=None):
        h, u, gg = saved[:3]
        keep = saved[3] if len(saved) > 3 else None          # FeedForward dropout: the keep mask of the forward (gg is the DROPPED gate output)
        W = FFInner.weights(meta['cache'], p)
        w1, w2 = p
        FP, FFI = W['FP'], W['FFI']
        if isinstance(dy, K.G16):
            # fp16-gradient backward: dy = fp16(S dy); every product on the fp16 MFMA against the fp16 copies the forward left
            s2 = dy.s2
            du = K.gemm_nt_geglu_bwd16(dy.t, W['w2T_16'], u.hi, FP)
            dw2 = torch.empty_like(w2)
            K.gemm_tn16(dy.t, gg.f16, dw2, s2, N2=FFI)
            dh = K.G16(K.gemm_nt_f16ops(du, W['w1T_16'], out_f16=True), s2)
            dw1p = torch.empty((2 * FP, w1.shape[1]), dtype=torch.float32, device=w1.device)
            K.gemm_tn16(du, h.f16, dw1p, s2)
            d = K.geglu_deinterleave(dw1p, FP, dim=0)
            return dh, None, [torch.cat((d[:FFI], d[FP:FP + FFI]), 0), dw2]
        if keep is not None:
            assert not isinstance(dy, K.G16)
            dgg = K.gemm_nt(dy, W['w2T'], out_bf16=True)
            dgd = K.empty_bf(tuple(dgg.hi.shape), dgg.hi.device)
            K.cast_pad(_ff_drop(_bf_val(dgg), keep, float(meta['drop_p'])).contiguous(), dgd)
            du = K.geglu_bwd(u, dgd, FP, interleaved=True)
        elif FUSE_GEGLU_BWD:
            du = K.gemm_nt_geglu_bwd(dy, W['w2T'], u, FP)      # dgg = dy W2 and the gate's backward in one pass
        else:
            du = K.geglu_bwd(u, K.gemm_nt(dy, W['w2T'], out_bf16=True), FP, interleaved=True)
        dw2 = torch.empty_like(w2)
        K.gemm_tn(dy, gg, dw2, N2=FFI)
        dh = K.gemm_nt(du, W['w1T'], out_bf16=_fast_bwd())
        sh = meta.get('shift')
        # one wgrad GEMM over the padded, interleaved [8 values | 8 gates | ...] row order, then back to the parameter's layout
        dw1p = torch.empty((2 * FP, w1.shape[1]), dtype=torch.float32, device=w1.device)

        def wgrad():
            K.gemm_tn(du, h, dw1p, shift=sh)
            d = K.geglu_deinterleave(dw1p, FP, dim=0)
            return torch.cat((d[:FFI], d[FP:FP + FFI]), 0)
        dw1 = wgrad()
        return dh, None, [dw1, dw2]


def _bf_val(t):
    """fp32 value of a BF pair"""
    return t.hi.float() if t.lo is None else t.hi.float() + t.lo.float()


def _bf_put(dst, index, value):
    """dst[index] = value (fp32) for a BF pair, rows selected by `index`"""
    hi = value.to(torch.bfloat16)
    dst.hi[index] = hi
    if dst.lo is not None:
        dst.lo[index] = (value - hi.float()).to(torch.bfloat16)


def _to_bf(value):
    out = K.empty_bf(tuple(value.shape), value.device)
    K.cast_pad(value.contiguous(), out)
    return out


class XC2Inner:
    """to_q(x), to_kv(sketch context) -> SparseCross2DNA core -> to_out (np.py:761-901).
    params: null_k, null_v, talking_heads.w, to_q.w, to_kv.w, to_out.w  (the layout of XInner)
    The windowed queries run on the 3DNA kernels pointed at the context (amdnuwa_cross2dna_*); the single <bos> query of every
    sample -- full attention over all context tokens, no talking heads (np.py:813-830) -- is B rows of glue arithmetic on torch ops
    between the kernels, forward and backward."""
    nparams = 6

    @staticmethod
    def _null(p, lo):
        nk, nv = p[0], p[1]
        mk = lambda t: _to_bf(t.detach().reshape(1, -1).float())
        a, b = mk(nk), mk(nv)
        return K.BF(a.hi.reshape(-1), None if a.lo is None else a.lo.reshape(-1)), K.BF(b.hi.reshape(-1), None if b.lo is None else b.lo.reshape(-1))

    @staticmethod
    def _bos_scores(q0, kf, nkf, keep, scale):
        """q0 [B,h,d], kf [B,T,h,d], nkf [h,d] -> softmax over [null | context] per (sample, head): [B,h,T+1]"""
        s = torch.cat((torch.einsum('bhd,hd->bh', q0, nkf)[..., None], torch.einsum('bhd,bthd->bht', q0, kf)), dim=-1) * scale
        if keep is not None:
            s = s.masked_fill(~torch.nn.functional.pad(keep, (1, 0), value=True)[:, None, :], -torch.finfo(torch.float32).max)
        return s.softmax(dim=-1)

    @staticmethod
    def fwd(h, p, meta):
        W = XInner.weights(meta['cache'], p)
        g, T = meta['geom'], meta['ctx_T']
        heads, dh = g.heads, g.dim_head
        inner = heads * dh
        wth2 = p[2].detach().reshape(heads, heads).contiguous()
        q = K.gemm_nt(h, W['q'], out_bf16=True)
        kv = K.gemm_nt(meta['ctx_bf'], W['kv'], out_bf16=True)
        nk, nv = XC2Inner._null(p, q.lo is not None)
        o = K.cross2dna_fwd(g, q, kv, nk, nv, meta['mask_u8'], wth2, T)
        # <bos> query rows
        rows0 = torch.arange(g.B, device=q.hi.device) * g.ntok
        q0 = _bf_val(K.BF(q.hi[rows0], None if q.lo is None else q.lo[rows0])).reshape(g.B, heads, dh)
        kvf = _bf_val(kv).reshape(g.B, T, 2, heads, dh)
        keep = meta['mask_u8'].bool() if meta['mask_u8'] is not None else None
        P0 = XC2Inner._bos_scores(q0, kvf[:, :, 0], _bf_val(nk).reshape(heads, dh), keep, g.scale)
        o0 = P0[..., :1] * _bf_val(nv).reshape(1, heads, dh) + torch.einsum('bht,bthd->bhd', P0[..., 1:], kvf[:, :, 1])
        _bf_put(o, rows0, o0.reshape(g.B, inner))
        y = K.gemm_nt(o, W['out'], out_bf16=_fast())
        return y, _sv(h, q, kv, nk, nv, o, P0)

    @staticmethod
    def bwd(saved, dy, p, meta, need_dbias=False, dy_f32=None):
        h, q, kv, nk, nv, o, P0 = saved
        W = XInner.weights(meta['cache'], p)
        nkp, nvp, wth, wq, wkv, wo = p
        g, T = meta['geom'], meta['ctx_T']
        heads, dh = g.heads, g.dim_head
        inner = heads * dh
        wth2 = wth.detach().reshape(heads, heads).contiguous()
        d_o = K.gemm_nt(dy, W['outT'], out_bf16=True)
        dwo = torch.empty_like(wo)
        K.gemm_tn(dy, o, dwo)
        dq, dkv, dnk, dnv, dwth = K.cross2dna_bwd(g, q, kv, nk, nv, meta['mask_u8'], wth2, d_o, T)
        # <bos> query rows: softmax attention backward on B rows
        rows0 = torch.arange(g.B, device=q.hi.device) * g.ntok
        sub = lambda t: _bf_val(K.BF(t.hi[rows0], None if t.lo is None else t.lo[rows0])).reshape(g.B, heads, dh)
        q0, dO0 = sub(q), sub(d_o)
        kvf = _bf_val(kv).reshape(g.B, T, 2, heads, dh)
        nkf, nvf = _bf_val(nk).reshape(heads, dh), _bf_val(nv).reshape(heads, dh)
        dP = torch.cat((torch.einsum('bhd,hd->bh', dO0, nvf)[..., None], torch.einsum('bhd,bthd->bht', dO0, kvf[:, :, 1])), dim=-1)
        dS = P0 * (dP - (P0 * dP).sum(-1, keepdim=True)) * g.scale
        dq0 = dS[..., :1] * nkf[None] + torch.einsum('bht,bthd->bhd', dS[..., 1:], kvf[:, :, 0])
        _bf_put(dq, rows0, dq0.reshape(g.B, inner))
        # the <bos> query's share of dK / dV is a separate (small) term: it goes through the two products below as its own operand instead
        # of being added into the kernel's bf16 dkv and rounded a second time
        dkv0 = _to_bf(torch.stack((torch.einsum('bht,bhd->bthd', dS[..., 1:], q0), torch.einsum('bht,bhd->bthd', P0[..., 1:], dO0)),
                                  dim=2).reshape(g.B * T, 2 * inner))
        dnk = dnk + torch.einsum('bh,bhd->hd', dS[..., 0], q0).reshape(-1)
        dnv = dnv + torch.einsum('bh,bhd->hd', P0[..., 0], dO0).reshape(-1)
        dh_ = K.gemm_nt(dq, W['qT'], out_bf16=_fast_bwd())
        dwq, dwkv = torch.empty_like(wq), torch.empty_like(wkv)
        K.gemm_tn(dq, h, dwq)
        K.gemm_tn(dkv, meta['ctx_bf'], dwkv)
        K.gemm_tn(dkv0, meta['ctx_bf'], dwkv, beta=1.0)
        dctx = K.gemm_nt(dkv, W['kvT']) + K.gemm_nt(dkv0, W['kvT'])
        return dh_, dctx, [dnk.reshape(nkp.shape), dnv.reshape(nvp.shape), dwth.reshape(wth.shape), dwq, dwkv, dwo]


INNERS = {'s3': S3Inner, 'xattn': XInner, 'ff': FFInner, 'xc2': XC2Inner}

FUSE_LINEAR_CE = os.environ.get('AMDNUWA_FUSE_LINEAR_CE', '1') != '0'   # to_logits + cross entropy without the fp32 logits (A/B switch)
# ... in 'bf16x3-fwd' (hi + lo logits): OFF by default.  The fused form needs the three-MFMA product twice (statistics, then dlogits -- the
# latter on one fp16 MFMA) where the unfused one writes the fp32 logits once and reads them once: measured 532.5-534.1 against 530.2-530.7 ms
# per step at b = 128, for 9 GB less peak memory (243 -> 234 GB).  AMDNUWA_FUSE_LINEAR_CE_X3 = 1 turns it on -- a trainer that runs the WHOLE
# reference step at b = 128 wants it: with tokenizer, text encoder and optimiser state next to the decoder the unfused step sits at 244 GiB of
# 288 and ran 708-919 ms depending on how often the allocator had to retry; fused it runs 701-713 ms at 235 GiB (tools/full_step.py turns it on
# from b = 112).  'auto' = fused when the fp32 logits + bf16 dlogits would push the device past AMDNUWA_FUSE_LINEAR_CE_X3_MEM_FRAC of its memory.
_F = os.environ.get('AMDNUWA_FUSE_LINEAR_CE_X3', '0')
FUSE_LINEAR_CE_X3 = True if _F == '1' else ('auto' if _F == 'auto' else False)
FUSE_LINEAR_CE_X3_MEM_FRAC = float(os.environ.get('AMDNUWA_FUSE_LINEAR_CE_X3_MEM_FRAC', '0.82'))


def _fuse_ce_x3(rows, classes, device):
    if FUSE_LINEAR_CE_X3 != 'auto':
        return bool(FUSE_LINEAR_CE_X3)
    if device.type != 'cuda':
        return False
    need = rows * classes * (4 + 2)             # fp32 logits + bf16 dlogits
    total = torch.cuda.get_device_properties(device).total_memory
    return torch.cuda.memory_allocated(device) + need > FUSE_LINEAR_CE_X3_MEM_FRAC * total


FUSE_GEGLU_BWD = os.environ.get('AMDNUWA_FUSE_GEGLU_BWD', '1') != '0'   # gate backward inside the dgg GEMM epilogue (A/B switch)
CHAIN_BWD = os.environ.get('AMDNUWA_CHAIN_BWD', '1') != '0'      # chain the LayerNorm backwards across block boundaries (A/B switch)
def _fast():
    """fast bf16 mode: the GEMMs that feed a LayerNorm (to_out / FF w2 outputs, dgrad outputs) write bf16 and the LN kernels
    read bf16 -- half the epilogue and LN traffic.  Parity mode (bf16x3) keeps those tensors in fp32."""
    return K.fast_io()


def _fast_bwd():
    """dgrad GEMM outputs that feed a LayerNorm backward are bf16 unless the BACKWARD itself runs 3-MFMA products ('bf16x3')"""
    return K.get_precision() != 'bf16x3'


def _sv(*ts):
    """what an inner stage keeps for its backward: in the mixed mode ('bf16x3-fwd') the hi parts only -- the lo parts die with the
    forward of the block"""
    return tuple(K.hi_only(t) for t in ts) if K.mixed() else ts


def _as_f32(t):
    return t.hi.float() if isinstance(t, K.BF) else t


GRAD_SCALE_LOG2 = float(os.environ.get('AMDNUWA_GRAD_SCALE_LOG2', '-4'))      # S * max|g| in [2^x, 2^(x+1))


def _grad_scale(g2):
    """device tensor {S, 1 / S} for the fp16 gradients of one backward pass: S = the power of two with S * max|g| in [2^-4, 2^-3), g = the
    residual-stream gradient entering the first block of the pass (kernels.G16).  No host synchronisation.
    Why that low (round 5 used [8, 16)): the gradients INSIDE the blocks are larger than the stream's -- a post-norm backward multiplies by
    w / std(y), and the cross-attention block's y is small -- and they grow towards layer 0: on the random-init cfg-3 stack (24 layers)
    |S g| of the cross-attention blocks' dy climbs from 1.2e3 at the top to beyond 65504 at layers 0 and 1 with [8, 16) (tools/grad_range.py,
    profiles/r06l_grad_range.txt: saturation counter 36).  Seven bits lower the largest tensor peaks at 2^9 and the smallest (|S g| max 0.1)
    still keeps 8 significant bits -- bf16's -- down to 2^-12 of its own maximum (fp16 is normal to 6.1e-5, subnormal to 6e-8)."""
    a = g2.detach().abs().amax().float()
    ok = torch.isfinite(a) & (a > 0)
    e = torch.floor(torch.log2(torch.where(ok, a, torch.ones_like(a))))
    S = torch.where(ok, torch.exp2((GRAD_SCALE_LOG2 - e).clamp(-60.0, 60.0)), torch.ones_like(a))
    return torch.stack((S, 1.0 / S)).contiguous()


def _block_bwd16(kind, R, D, p, meta):
    """does this block run the fp16-gradient backward (and so keep only the fp16 copy of its LayerNorm input)?"""
    if not K.bwd_f16() or meta.get('shift_unfused'):
        return False
    if kind == 'ff':
        return not meta.get('drop_p') and FFInner.bwd16_ok(R, D, _ru(p[1].shape[1], 32), p[1].shape[1], (p[0], p[1]))
    if kind == 's3':
        return S3Inner.bwd16_ok(R, D, p[0].shape[0], meta['geom'], (p[0], p[1]), p[3], len(p) > 5)
    if kind == 'xattn':
        return XInner.bwd16_ok(R, D, p[3].shape[0], meta['xgeom'], meta, (p[3], p[5]))
    return False


_CTX_CAST = [None, -1, None]      # (weak reference to the context tensor, its version, its BF copy)


def _ctx_to_bf(context):
    """context fp32 [B, T, D] -> BF [B*T, D].  Every cross-attention block of a stack receives the SAME context tensor: its cast (27 us at
    cfg 3, b = 128) runs once per tensor object and version, not once per layer (the copy is read-only: kv projection, its weight gradient)"""
    ref = _CTX_CAST[0]
    if ref is not None and ref() is context and _CTX_CAST[1] == context._version:
        return _CTX_CAST[2]
    B, T, D = context.shape
    out = K.empty_bf((B * T, D), context.device)
    K.cast_pad(context.detach().reshape(B * T, D), out)
    try:
        _CTX_CAST[0], _CTX_CAST[1], _CTX_CAST[2] = weakref.ref(context), context._version, out
    except TypeError:
        _CTX_CAST[0] = None
    return out


# =================================================================================================
# fused decoder sub-block:  x_out = x + postLN(inner(shift(preLN(x))))
# =================================================================================================

class SandwichBlockFn(Function):
    """x_out = (resid if given else x) + postLN(inner(shift(preLN(x)))).
    args: x [B, n, D] fp32, resid (or None), context (or None), meta dict, pre_w, pre_b, post_w, post_b, *inner params"""

    @staticmethod
    @_in_phase('fwd')
    def forward(ctx, x, resid, context, meta, pre_w, pre_b, post_w, post_b, *p):
        inner = INNERS[meta['kind']]
        B, n, D = x.shape
        x2 = x.detach().contiguous().reshape(B * n, D)
        r2_ = x2 if resid is None else resid.detach().contiguous().reshape(B * n, D)
        meta = dict(meta)
        if meta['kind'] in ('xattn', 'xc2') and not meta.get('self_kv'):
            meta['ctx_bf'] = _ctx_to_bf(context)
        # the token shift is folded into the pre-LN's STORE: h = shift(LN(x)) is what the forward GEMM and the weight-gradient
        # GEMM consume (plain loaders); only the pre-LN backward still reads its incoming gradient through the inverse shift
        sh = meta.get('shift')
        # block chaining (Transformer.forward_layers): the previous block's post-norm kernel may already have produced this
        # block's h = shift(LN(x)) while the new stream row was in its registers, and this block does the same for the next
        hin, nxt, hout = meta.pop('handoff_in', None), meta.pop('next_pre', None), meta.pop('handoff_out', None)
        ctx.prev_ctx = None
        # h as a bf16 + fp16 copy pair when this block's first GEMM runs fp16 operands ('bf16x3-fwd': FeedForward, the 3DNA projection)
        want16 = (meta['kind'] == 'ff' and FFInner.f16_ok(B * n, D, _ru(p[1].shape[1], 32), (p[0], p[1]))) or \
                 (meta['kind'] == 's3' and S3Inner.f16_proj_ok(B * n, D, p[0].shape[0], meta['geom'], (p[0], p[1]))) or \
                 (meta['kind'] == 'xattn' and XInner.f16x2_ok(B * n, D, p[3].shape[0], meta['xgeom'], meta, (p[3], p[5])))
        # (blocks of a reversible stack -- a separate residual input -- keep the bf16 backward: they are not chained, so every one of them would
        #  take its own gradient scale: one amax pass per block, -2.7 % on cfg 4)
        bw16 = bool(want16) and resid is None and _block_bwd16(meta['kind'], B * n, D, p, meta)
        if hin is not None and hin.get('ptr') == x.data_ptr() and hin.get('ver') == x._version and hin.get('shift') == sh \
                and resid is None and K.bf_rows_cols(hin['h'])[:2] == (B * n, D) and (hin['h'].hi is not None or bw16):
            h, m1, r1 = hin['h'], hin['m1'], hin['r1']
            ctx.prev_ctx = hin.get('ctx')          # the backward chains the two LayerNorm backwards of this boundary too
        else:
            h, m1, r1, _ = K.ln_fwd(x2, pre_w.detach(), pre_b.detach(), shift=sh, f16='only' if bw16 else want16)
        ctx.shift = sh
        if sh is not None:
            meta['shift'] = None
        meta['bwd16'] = ctx.bw16 = bw16 and h.f16 is not None
        y, saved = inner.fwd(h, p, meta)
        if nxt is not None and hout is not None:
            # the next block's first GEMM runs fp16 operands: FeedForward (nxt[3] = ('ff', inner width)) or the 3DNA projection (('s3', inner, geom))
            nk = nxt[3] if len(nxt) > 3 else None
            nxt16 = nk is not None and ((nk[0] == 'ff' and FFInner.f16_ok(B * n, D, _ru(nk[1], 32), nk[2])) or
                                        (nk[0] == 's3' and S3Inner.f16_proj_ok(B * n, D, nk[1], nk[2], nk[3])) or
                                        (nk[0] == 'x' and XInner.f16x2_ok(B * n, D, nk[1], nk[2], nk[3], nk[4])))
            if nxt16 and ((nk[0] == 'ff' and FFInner.bwd16_ok(B * n, D, _ru(nk[1], 32), nk[1], nk[2])) or
                          (nk[0] == 's3' and len(nk) > 5 and S3Inner.bwd16_ok(B * n, D, nk[1], nk[2], nk[3], nk[4], nk[5])) or
                          (nk[0] == 'x' and XInner.bwd16_ok(B * n, D, nk[1], nk[2], nk[3], nk[4]))):
                nxt16 = 'only'                    # the next block keeps ONE (fp16) copy of its LayerNorm input: fp16-gradient backward
            xo, m2, r2, hn, mn, rn = K.ln_post_pre_fwd(y, r2_, post_w.detach(), post_b.detach(), nxt[0].detach(),
                                                       nxt[1].detach(), next_shift=nxt[2], next_f16=nxt16)
            hout.update(h=hn, m1=mn, r1=rn, ptr=xo.data_ptr(), ver=xo._version, shift=nxt[2], ctx=ctx if CHAIN_BWD else None)
        else:
            xo, m2, r2 = K.ln_fwd(y, post_w.detach(), post_b.detach(), resid=r2_, minus=bool(meta.get('resid_minus')))
        ctx.meta, ctx.inner_saved, ctx.p = meta, saved, p
        ctx.has_ctx = context is not None
        ctx.has_resid = resid is not None
        ctx.y_bf = isinstance(y, K.BF)
        ctx.save_for_backward(x2, y.hi if ctx.y_bf else y, m1, r1, m2, r2, pre_w, post_w)
        ctx.shape = (B, n, D)
        return xo.reshape(B, n, D)

    @staticmethod
    @_in_phase('bwd')
    def backward(ctx, g):
        x2, y, m1, r1, m2, r2, pre_w, post_w = ctx.saved_tensors
        if ctx.y_bf:
            y = K.BF(y, None)
        B, n, D = ctx.shape
        meta, p = ctx.meta, ctx.p
        inner = INNERS[meta['kind']]
        g2 = g.contiguous().reshape(B * n, D)
        want_bias = meta['kind'] == 's3'
        ho, ctx.bwd_handoff = getattr(ctx, 'bwd_handoff', None), None
        bw16 = getattr(ctx, 'bw16', False)
        s2 = ho.get('s2') if ho is not None else None          # the gradient scale of this backward pass (fp16-gradient blocks)
        # (only where this block or the one its chained LayerNorm backward feeds runs on fp16 gradients: a reversible stack -- no such block --
        #  paid one amax pass over the stream per block for a scale nobody read: cfg 4 -4 %)
        if s2 is None and K.bwd_f16() and (bw16 or getattr(ctx.prev_ctx, 'bw16', False)):
            s2 = _grad_scale(g2)
        if ho is not None and ho['ptr'] == g2.data_ptr() and ho['ver'] == g2._version and isinstance(ho['dy'], K.G16) == bw16:   # the next block's backward already ran this post-norm backward
            dy, dpost_w, dpost_b, dsum = ho['dy'], ho['dw'], ho['db'], ho['dsum']
        else:
            dy, dpost_w, dpost_b, dsum = K.ln_bwd(g2, y, m2, r2, post_w.detach(), to_bf=not bw16, want_dsum=want_bias,
                                                  to_f16=s2 if bw16 else None)
        meta = dict(meta)
        dh, dctx, grads = inner.bwd(ctx.inner_saved, dy, p, meta, need_dbias=False)
        if want_bias:
            grads[4] = dsum                      # to_out.bias grad = column sums of d(to_out output)
        prev = ctx.prev_ctx
        p16 = prev is not None and getattr(prev, 'bw16', False) and s2 is not None
        if prev is not None and not ctx.has_resid and (isinstance(dh, (K.BF, K.G16)) or not prev.y_bf) and (not isinstance(dh, K.BF) or dh.lo is None) \
                and not (isinstance(dh, K.G16) and prev.y_bf):
            # pre-norm backward of this block + post-norm backward of the previous block on the same gradient row
            _, py, _, _, pm2, pr2, _, ppost_w = prev.saved_tensors
            dx, dpre_w, dpre_b, dyp, dwp, dbp, dsp = K.ln_bwd_chain(
                dh, x2, m1, r1, pre_w.detach(), g2, K.BF(py, None) if prev.y_bf else py, pm2, pr2, ppost_w.detach(),
                shift=ctx.shift, want_dsum=prev.meta['kind'] == 's3', out_f16=s2 if p16 else None)
            prev.bwd_handoff = dict(ptr=dx.data_ptr(), ver=dx._version, dy=dyp, dw=dwp, db=dbp, dsum=dsp, s2=s2)
        else:
            # (dx_add: a gradient the caller would add to dx anyway -- the reversible reconstruction's dy1 + dL/dy1 -- rides in as the pre-norm
            #  backward's accumulator, as the residual-stream gradient does on the plain stack: no separate add pass over the stream)
            add = getattr(ctx, 'dx_add', None)
            ctx.dx_add = None
            if add is not None:
                add = add.detach().contiguous().reshape(B * n, D)
            dx, dpre_w, dpre_b, _ = K.ln_bwd(dh, x2, m1, r1, pre_w.detach(), dres=(add if ctx.has_resid else g2), shift=ctx.shift)
        ctx.prev_ctx = None
        dcontext = None
        if ctx.has_ctx:
            T = meta['ctx_T'] if meta['kind'] == 'xc2' else meta['xgeom'].T
            dcontext = dctx.reshape(B, T, D)
        ctx.inner_saved = None
        dresid = g if ctx.has_resid else None
        return (dx.reshape(B, n, D), dresid, dcontext, None, dpre_w, dpre_b, dpost_w, dpost_b, *grads)


# =================================================================================================
# standalone inner module call:  y = inner(x)  (fp32 in / fp32 out), e.g. Sparse3DNA(x) on its own
# =================================================================================================

class InnerFn(Function):
    @staticmethod
    @_in_phase('fwd')
    def forward(ctx, x, context, meta, *p):
        inner = INNERS[meta['kind']]
        B, n, D = x.shape
        x2 = x.detach().contiguous().reshape(B * n, D)
        meta = dict(meta)
        if meta['kind'] in ('xattn', 'xc2') and not meta.get('self_kv'):
            meta['ctx_bf'] = _ctx_to_bf(context)
        h = K.empty_bf((B * n, D), x.device)
        K.cast_pad(x2, h)
        y, saved = inner.fwd(h, p, meta)
        ctx.meta, ctx.inner_saved, ctx.p = meta, saved, p
        ctx.has_ctx = context is not None
        ctx.shape = (B, n, D)
        y = _as_f32(y)
        return y.reshape(B, n, y.shape[-1])

    @staticmethod
    @_in_phase('bwd')
    def backward(ctx, g):
        B, n, D = ctx.shape
        meta, p = ctx.meta, ctx.p
        inner = INNERS[meta['kind']]
        g2 = g.contiguous().reshape(B * n, -1)
        dy = K.empty_bf(tuple(g2.shape), g.device)
        K.cast_pad(g2, dy)
        meta = dict(meta)
        dh, dctx, grads = inner.bwd(ctx.inner_saved, dy, p, meta, need_dbias=True, dy_f32=g2)
        dcontext = dctx.reshape(B, meta['ctx_T'] if meta['kind'] == 'xc2' else meta['xgeom'].T, -1) if ctx.has_ctx else None
        ctx.inner_saved = None
        return (_as_f32(dh).reshape(B, n, D), dcontext, None, *grads)


# =================================================================================================
# LayerNorm / StableLayerNorm as standalone nodes (fp32 in -> fp32 out)
# =================================================================================================

class LayerNormFn(Function):
    @staticmethod
    @_in_phase('fwd')
    def forward(ctx, x, w, b, stable):
        shp = x.shape
        x2 = x.detach().contiguous().reshape(-1, shp[-1])
        zero = torch.zeros_like(x2)
        if stable:
            raise RuntimeError('use StableLNLogitsFn / stable_ln_bf')
        out, m, r = K.ln_fwd(x2, w.detach(), b.detach(), resid=zero)
        ctx.save_for_backward(x2, m, r, w)
        ctx.shp = shp
        return out.reshape(shp)

    @staticmethod
    @_in_phase('bwd')
    def backward(ctx, g):
        x2, m, r, w = ctx.saved_tensors
        g2 = g.contiguous().reshape(x2.shape)
        dx, dw, db, _ = K.ln_bwd(g2, x2, m, r, w.detach())
        return dx.reshape(ctx.shp), dw, db, None


class StableLNFn(Function):
    """StableLayerNorm (np.py:88-95) -> fp32 output (value of the bf16 hi[/lo] pair the logits GEMM consumes is
    produced separately in LogitsFn; standalone use returns fp32)."""

    @staticmethod
    @_in_phase('fwd')
    def forward(ctx, x, w, b):
        shp = x.shape
        x2 = x.detach().contiguous().reshape(-1, shp[-1])
        out, m, r, ia = K.ln_fwd(x2, w.detach(), b.detach(), stable=True)
        ctx.save_for_backward(x2, m, r, ia, w)
        ctx.shp = shp
        y = out.hi.float()
        if out.lo is not None:
            y = y + out.lo.float()
        return y.reshape(shp)

    @staticmethod
    @_in_phase('bwd')
    def backward(ctx, g):
        x2, m, r, ia, w = ctx.saved_tensors
        g2 = g.contiguous().reshape(x2.shape)
        dx, dw, db, _ = K.ln_bwd(g2, x2, m, r, w.detach(), inv_amax=ia)
        return dx.reshape(ctx.shp), dw, db


# =================================================================================================
# final StableLayerNorm + to_logits (+ cross entropy)   np.py:1182, 1958-1963
# =================================================================================================

def _logits_f16x2(W, wl, R, D):
    """'bf16x3-fwd' with the two-MFMA switch on: to_logits takes the final LayerNorm output as ONE fp16 value and the weight as an fp16
    hi + lo pair (built once per weight version; None = outside the fp16 range)"""
    if not K.proj_f16x2('l'):
        return False
    if 'w_16' not in W:
        W['w_16'] = K.f16_pair(wl) if f16_weights_ok(wl) else None
    return W['w_16'] is not None and K.gemm_nt_f16x2_ok(R, wl.shape[0], D, out_bf16=False)


class LogitsFn(Function):
    """x [B, n, D] -> logits [B, n, C] fp32 (StableLayerNorm then Linear without bias)"""

    @staticmethod
    @_in_phase('fwd')
    def forward(ctx, x, nw, nb, wl, cache):
        B, n, D = x.shape
        x2 = x.detach().contiguous().reshape(B * n, D)
        W = cache.get('logits', (wl,), lambda: dict(w=_cast(wl), wT=_cast_t(wl)))
        lx2 = _logits_f16x2(W, wl, B * n, D)
        hn, m, r, ia = K.ln_fwd(x2, nw.detach(), nb.detach(), stable=True, f16=lx2)
        logits = K.gemm_nt_f16x2(hn.f16, W['w_16']) if lx2 else K.gemm_nt(hn, W['w'])
        ctx.save_for_backward(x2, m, r, ia, nw, wl)
        ctx.hn, ctx.W, ctx.shape = _sv(hn)[0], W, (B, n, D)
        return logits.reshape(B, n, -1)

    @staticmethod
    @_in_phase('bwd')
    def backward(ctx, g):
        x2, m, r, ia, nw, wl = ctx.saved_tensors
        B, n, D = ctx.shape
        g2 = g.contiguous().reshape(B * n, -1)
        dl = K.empty_bf(tuple(g2.shape), g.device)
        K.cast_pad(g2, dl)
        dhn = K.gemm_nt(dl, ctx.W['wT'])
        dwl = torch.empty_like(wl)
        K.gemm_tn(dl, ctx.hn, dwl)
        dx, dnw, dnb, _ = K.ln_bwd(dhn, x2, m, r, nw.detach(), inv_amax=ia)
        ctx.hn = None
        return dx.reshape(B, n, D), dnw, dnb, dwl, None


class LogitsLossFn(Function):
    """x [B, n, D], targets [B, n] -> scalar mean cross-entropy.  dlogits is produced in the forward
    (bf16 hi[/lo], already divided by the number of targets) so the fp32 logits are read only once."""

    @staticmethod
    @_in_phase('fwd')
    def forward(ctx, x, targets, nw, nb, wl, cache):
        B, n, D = x.shape
        x2 = x.detach().contiguous().reshape(B * n, D)
        W = cache.get('logits', (wl,), lambda: dict(w=_cast(wl), wT=_cast_t(wl)))
        fuse_x3 = K.mixed() and _fuse_ce_x3(B * n, wl.shape[0], x.device)
        if fuse_x3 and 'w16' not in W:       # fp16 copy for the dlogits pass of the fused cross entropy (None: outside the fp16 range)
            W['w16'] = wl.detach().to(torch.float16).contiguous() if f16_weights_ok(wl) else None
        use_fused = fuse_x3 if K.mixed() else FUSE_LINEAR_CE
        lx2 = not use_fused and _logits_f16x2(W, wl, B * n, D)
        hn, m, r, ia = K.ln_fwd(x2, nw.detach(), nb.detach(), stable=True, f16=lx2)
        if targets.dtype != torch.int64:
            raise TypeError(f'cross-entropy targets must be int64 token ids, got {targets.dtype}')
        if targets.numel() != B * n:
            raise ValueError(f'{targets.numel()} targets for {B * n} logit rows')
        t = targets.contiguous().reshape(-1)            # ids outside [0, C) give a NaN loss (the kernel never reads out of bounds)
        want_grad = any(ctx.needs_input_grad)
        fused = K.linear_ce(hn, W['w'], t, 1.0 / (B * n), want_grad=want_grad, w16=W.get('w16')) if use_fused else None
        if fused is not None:                           # 'bf16' (and 'bf16x3-fwd' on request): logits produced twice inside the GEMM ring, never written (np.py:1958-1963)
            loss, dl = fused
        else:
            logits = K.gemm_nt_f16x2(hn.f16, W['w_16']) if lx2 else K.gemm_nt(hn, W['w'])
            loss, dl = K.ce_fwd(logits, t, 1.0 / (B * n), want_grad=want_grad, lo=False if K.mixed() else None)
            del logits
        ctx.save_for_backward(x2, m, r, ia, nw, wl)
        ctx.hn, ctx.dl, ctx.W, ctx.shape = _sv(hn)[0], dl, W, (B, n, D)
        return loss

    @staticmethod
    @_in_phase('bwd')
    def backward(ctx, g):
        x2, m, r, ia, nw, wl = ctx.saved_tensors
        B, n, D = ctx.shape
        assert ctx.dl.hi is not None
        dhn = K.gemm_nt(ctx.dl, ctx.W['wT'])
        dwl = torch.empty_like(wl)
        K.gemm_tn(ctx.dl, ctx.hn, dwl)
        gs = g.detach().reshape(1).float().contiguous()
        K.scale_by_device_scalar(dwl, gs)
        # the upstream gradient of the loss rides into the final norm's backward as a device scalar (it cost a read-modify-write pass over dhn)
        dx, dnw, dnb, _ = K.ln_bwd(dhn, x2, m, r, nw.detach(), inv_amax=ia, dy_scale2=torch.cat((gs, gs)))
        ctx.hn = ctx.dl = None
        return dx.reshape(B, n, D), None, dnw, dnb, dwl, None


# =================================================================================================
# embedding assemble  (np.py:1940-1944)
# =================================================================================================

class EmbedAssembleFn(Function):
    """ids [B, n-1] int64 -> x [B, n, D] fp32 = cat(bos, pos[:n-1] + frac_gradient(emb(ids)))"""

    @staticmethod
    @_in_phase('fwd')
    def forward(ctx, ids, W, ax1, ax2, ax3, bos, video_shape, frac):
        B, n1 = ids.shape
        ntok = n1 + 1
        F, H, Wd = video_shape
        ids_c = ids.contiguous()
        x = K.embed_fwd(ids_c, W.detach(), ax1.detach(), ax2.detach(), ax3.detach(), bos.detach(), B, ntok, H, Wd, frac)
        ctx.save_for_backward(ids_c)
        ctx.meta = (B, ntok, F, H, Wd, frac, W.shape, ax1.shape, ax2.shape, ax3.shape, bos.shape)
        return x.reshape(B, ntok, -1)

    @staticmethod
    @_in_phase('bwd')
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        B, ntok, F, H, Wd, frac, ws, s1, s2, s3, sb = ctx.meta
        dev = g.device
        g2 = g.contiguous().reshape(B * ntok, -1)
        dW = torch.zeros(ws, dtype=torch.float32, device=dev)
        d1, d2, d3 = (torch.zeros(s, dtype=torch.float32, device=dev) for s in (s1, s2, s3))
        db = torch.zeros(sb, dtype=torch.float32, device=dev)
        K.embed_bwd(ids, g2, dW, d1, d2, d3, db, B, ntok, F, H, Wd, frac)
        return None, dW, d1, d2, d3, db, None, None
