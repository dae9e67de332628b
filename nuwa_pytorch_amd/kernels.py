"""Thin torch-tensor wrappers over the C-ABI (one function per entry point of include/amdnuwa.h).
Tensors only provide device memory (torch caching allocator) and the current HIP stream; all
arithmetic happens inside libamdnuwa."""
import ctypes as C
import os
import threading
from collections import namedtuple

import torch

from . import _lib
from ._lib import GemmDesc, S3Geom, XGeom, XKV, check

# bf16 hi part + optional bf16 residual `lo` (hi + lo = the value to ~16 bits: the operand form of the 3-MFMA products) + optional
# `f16` = the fp16 rendering of the same value (the operand form of the fp16 attention cores of the 'bf16x3-fwd' mode)
BF = namedtuple('BF', ['hi', 'lo', 'f16'], defaults=(None,))

DEFAULT_PRECISION = 'bf16x3-fwd'          # the mode that meets the north star's 1e-3 logits bound (and the one bench.py reports)
_PRECISION = os.environ.get('AMDNUWA_PRECISION', DEFAULT_PRECISION)
_TIMER = {'on': False, 'flops': 0.0, 'tags': [], 'fam': {}}
MODES = ('bf16', 'bf16x3', 'bf16x3-fwd')
_TL = threading.local()
if _PRECISION not in MODES:
    raise ValueError(f'AMDNUWA_PRECISION={_PRECISION!r}: expected one of {MODES}')


def set_precision(mode):
    """'bf16'      : bf16 MFMA operands, fp32 accumulate / softmax / LayerNorm / residual stream (fastest; logits ~8e-3 of the fp32
                  reference at full cfg-3 depth: does NOT meet the 1e-3 bound).
    'bf16x3'    : every MFMA operand carried as a bf16 hi+lo pair, 3 MFMAs per product (~fp32 accuracy) in the forward AND the
                  backward (parity mode for gradients).
    'bf16x3-fwd': (package default) the cheapest forward arithmetic that keeps the full-depth logits within 1e-3 of the fp32
                  reference, placed per product by an error budget:
                    * the cross-attention kv projection and to_logits: bf16 hi + lo operand pairs, 3 MFMAs per product;
                    * to_out (both attention blocks) and the cross-attention q projection: TWO fp16 MFMAs per product -- the
                      activation as ONE fp16 value (the fp16 cores / the LayerNorm store hand it over), the weight as an fp16
                      hi + lo pair (exact to ~22 bits); set_proj_f16x2() moves products between this form and the 3-MFMA one;
                    * the Sparse3DNA q / k / v projection, FF1 (+ GEGLU gate) and FF2: SINGLE fp16 MFMAs on fp16 copies of the
                      LayerNorm outputs and of the weights (11 significand bits; weights outside fp16's range fall back to the
                      hi + lo form, activations saturate at +-65504);
                    * both attention cores: single fp16 MFMAs on fp16 q / k / v and fp16 probabilities, outputs as a bf16 copy (for
                      the backward) + an fp16 copy (to_out's operand; a bf16 hi + lo pair when to_out runs 3 MFMAs);
                  fp32 everywhere else (LayerNorm, softmax, residual stream, accumulators).  The backward runs single bf16 MFMAs on
                  the bf16 copies as in 'bf16' (gradients carry bf16-level error).  The fp16 parts can be switched off one by
                  one (set_cores_f16 / set_ff_f16 / set_qkv_f16 / set_proj_f16x2, env AMDNUWA_F16_CORES / _FF / _QKV / AMDNUWA_F16X2 = 0):
                  with all of them off the forward IS the 'bf16x3' forward, bit for bit.  The 16-bit second copies live only inside the forward of one block."""
    global _PRECISION
    if mode not in MODES:
        raise ValueError(mode)
    _PRECISION = mode


def get_precision():
    return _PRECISION


class phase:
    """marks the code of an autograd node as forward ('fwd') or backward ('bwd') work.  Only the mixed mode cares: its backward
    allocates and consumes hi-only operands.  Every Function.forward / backward of ops.py runs inside one (a recomputing
    backward -- the reversible stacks -- re-enters 'fwd' through the Function.forward calls it makes)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = getattr(_TL, 'phase', 'fwd')
        _TL.phase = self.name

    def __exit__(self, *exc):
        _TL.phase = self.prev


def in_backward():
    return getattr(_TL, 'phase', 'fwd') == 'bwd'


def want_lo():
    return _PRECISION == 'bf16x3' or (_PRECISION == 'bf16x3-fwd' and not in_backward())


def mixed():
    return _PRECISION == 'bf16x3-fwd'


def fast_io():
    """only the all-bf16 mode stores the GEMM outputs that feed a LayerNorm (and the dgrad outputs) as bf16"""
    return _PRECISION == 'bf16'


def hi_only(t):
    """the hi part of a BF pair (what the mixed mode keeps for its bf16 backward); anything else passes through"""
    if isinstance(t, BF) and t.hi is None:
        return t                                   # fp16-only form: the backward of the block reads the fp16 copy
    return BF(t.hi, None) if isinstance(t, BF) and (t.lo is not None or t.f16 is not None) else t


_CORES_F16 = os.environ.get('AMDNUWA_F16_CORES', '1') != '0'


def set_cores_f16(on):
    """'bf16x3-fwd' only: run the forward attention cores on single fp16 MFMAs (default) or, when off, on 3-MFMA bf16 hi + lo
    products like the projection GEMMs around them (A/B switch; both meet the 1e-3 logits bound)"""
    global _CORES_F16
    _CORES_F16 = bool(on)


def cores_f16():
    return _CORES_F16 and mixed()


# fp16 gradients in the backward of 'bf16x3-fwd' (round 5).  A block whose class is switched on keeps ONE 16-bit copy of each activation --
# the fp16 one its forward reads -- and runs its backward on fp16 MFMAs as well: gradients travel as fp16(S * value), S a power of two taken
# once per backward pass from the residual-stream gradient (ops._grad_scale), weight gradients leave the split-K reduction multiplied by
# 1 / S.  Classes: 'f' FeedForward, 's' Sparse3DNA, 'x' cross attention (env AMDNUWA_BWD_F16, '0' / '' = off).
DEFAULT_BWD_F16 = 'fsx'
G16 = namedtuple('G16', ['t', 's2'])        # t: fp16 tensor holding S * value; s2: device tensor {S, 1 / S}


def _bwd16_classes(v):
    if v in (True, '1', 'all'):
        return frozenset('fsx')
    if v in (False, None, '0', ''):
        return frozenset()
    assert set(v) <= set('fsx'), v
    return frozenset(v)


_BWD_F16 = _bwd16_classes(os.environ.get('AMDNUWA_BWD_F16', DEFAULT_BWD_F16))


def set_bwd_f16(on):
    """'bf16x3-fwd' only: which block classes run their backward on fp16 gradients and keep a single (fp16) copy of their activations"""
    global _BWD_F16
    _BWD_F16 = _bwd16_classes(on)


def bwd_f16(cls=None):
    return mixed() and (bool(_BWD_F16) if cls is None else cls in _BWD_F16)


def f16_sat_count(reset=True):
    """threads that handed a value beyond +-65504 to a saturating fp16 store since the last reset (synchronises the device)"""
    return int(_lib.lib().amdnuwa_f16_sat_count(1 if reset else 0))


def bf_rows_cols(t):
    """(rows, cols, device) of a 2-D BF operand in any of its forms"""
    x = t.hi if t.hi is not None else t.f16
    return x.shape[0], x.shape[1], x.device


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('nuwa_pytorch_amd kernels need CUDA(HIP) device tensors; there is no CPU fallback')


def empty_bf(shape, device, lo=None):
    lo = want_lo() if lo is None else lo
    return BF(torch.empty(shape, dtype=torch.bfloat16, device=device),
              torch.empty(shape, dtype=torch.bfloat16, device=device) if lo else None)


def zeros_bf(shape, device, lo=None):
    lo = want_lo() if lo is None else lo
    return BF(torch.zeros(shape, dtype=torch.bfloat16, device=device),
              torch.zeros(shape, dtype=torch.bfloat16, device=device) if lo else None)


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def view(b, rows=None, cols=None):
    """2-D sub-view of a BF pair"""
    r = rows if rows is not None else slice(None)
    c = cols if cols is not None else slice(None)
    return BF(b.hi[r, c], None if b.lo is None else b.lo[r, c])


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------

def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, 'GEMM operands must be 2-D with unit inner stride'
    return t.stride(0)


def timer_arm(on):
    _TIMER['on'] = bool(on)
    _TIMER['flops'] = 0.0
    _TIMER['issued'] = 0.0
    _TIMER['bytes'] = 0.0
    _TIMER['tags'] = []
    _TIMER['fam'] = {}
    _lib.lib().amdnuwa_timer_arm(1 if on else 0)


def timer_collect():
    """(ms, launches, algorithmic flops, algorithmic bytes) of the NT GEMM family + timer_families() for every bracketed family"""
    tags = _TIMER['tags']
    n = C.c_longlong(0)
    buf = (C.c_double * max(len(tags), 1))()
    check(_lib.lib().amdnuwa_timer_collect_each(buf, len(buf), C.byref(n)), 'timer_collect_each')
    assert n.value == len(tags), (n.value, len(tags))
    per = {}
    for t, ms in zip(tags, buf):
        e = per.setdefault(t, [0.0, 0])
        e[0] += ms
        e[1] += 1
    fam = _TIMER['fam']
    fam['gemm_nt'] = [_TIMER['flops'], _TIMER.get('bytes', 0.0)]
    _TIMER['families'] = {k: dict(ms=v[0], launches=v[1], flops=fam.get(k, [0.0, 0.0])[0], bytes=fam.get(k, [0.0, 0.0])[1]) for k, v in per.items()}
    _TIMER['tags'] = []
    g = per.get('gemm_nt', [0.0, 0])
    return g[0], g[1], _TIMER['flops'], _TIMER.get('bytes', 0.0)


def timer_families():
    """per family of the last timer_collect(): {'ms', 'launches', 'flops', 'bytes'} -- gemm_nt, xattn, s3, ln"""
    return _TIMER.get('families', {})


def _big_bytes(*objs):
    """bytes of the distinct large tensors among the arguments / results of a kernel wrapper: its algorithmic HBM traffic (every operand
    read once, every result written once)"""
    seen, tot = set(), 0

    def walk(o):
        nonlocal tot
        if isinstance(o, torch.Tensor):
            if o.is_cuda and o.numel() >= 65536 and o.data_ptr() not in seen:
                seen.add(o.data_ptr())
                tot += o.numel() * o.element_size()
        elif isinstance(o, (list, tuple)):
            for x in o:
                walk(x)
        elif isinstance(o, dict):
            for x in o.values():
                walk(x)
        elif isinstance(o, BF):
            walk((o.hi, o.lo, o.f16))
        elif hasattr(o, 't') and isinstance(getattr(o, 't'), torch.Tensor):
            walk(o.t)
    walk(objs)
    return float(tot)


def _family(name, work=None):
    """bench.py's per-family launch timer: brackets the wrapper's launches while the timer is armed (work(args, kwargs, result) ->
    (algorithmic flops, algorithmic bytes); default: no flops, the large tensors of the call)"""
    def deco(fn):
        def wrap(*a, **k):
            if not _TIMER['on']:
                return fn(*a, **k)
            L, st = _lib.lib(), _stream()
            _TIMER['tags'].append(name)
            L.amdnuwa_timer_begin(st)
            r = fn(*a, **k)
            L.amdnuwa_timer_end(st)
            fl, by = work(a, k, r) if work else (0.0, _big_bytes(a, k, r))
            e = _TIMER['fam'].setdefault(name, [0.0, 0.0])
            e[0] += fl
            e[1] += by
            return r
        wrap.__name__, wrap.__doc__ = fn.__name__, fn.__doc__
        return wrap
    return deco


def _x_work(kind):
    """algorithmic MFMA work of the cross-attention cores (U = one 2 n (T + 1) d h product): forward QK^T + P'V + head mix; backward query
    side S, dP', dq + two mixes + dW_th; the batched dK / dV products 2 U"""
    def w(a, k, r):
        g = a[0]
        U = 2.0 * g.B * g.n * (g.T + 1) * g.dim_head * g.heads
        mix = 2.0 * g.B * g.n * (g.T + 1) * g.heads * g.heads
        fl = {'fwd': 2 * U + mix, 'bwd': 3 * U + 3 * mix, 'kv': 2 * U}[kind]
        act = 2.0 * g.B * g.n * g.heads * g.dim_head                       # one 16-bit [B*n, inner] tensor
        by = {'fwd': 3 * act, 'bwd': 3 * act + 4.0 * g.B * g.heads * g.n * (g.T + 1), 'kv': 2 * act + 4.0 * g.B * g.heads * g.n * (g.T + 1)}[kind]
        return fl, by
    return w


def _s3_work(kind):
    """algorithmic HBM bytes of the Sparse3DNA cores: forward reads q, k, v and writes o; backward reads q, k, v, dO and writes dq, dk, dv"""
    def w(a, k, r):
        g = a[0]
        act = 2.0 * g.B * g.ntok * g.heads * g.dim_head
        J = g.kf * g.kh * g.kw + 1
        fl = 4.0 * g.B * g.ntok * J * g.heads * g.dim_head * (1 if kind == 'fwd' else 2.5)
        return fl, act * (4 if kind == 'fwd' else 8)
    return w


def timer_issued_flops():
    """FLOPs the MFMA pipe was actually asked for in the armed region (3 x the algorithmic 2MNK where operands travel as hi + lo pairs)"""
    return _TIMER.get('issued', 0.0)


def hilo_to_f16(t):
    """fp16 tensor = fp16(hi + lo) of a 2-D BF pair"""
    R, Cc = t.hi.shape
    out = torch.empty((R, Cc), dtype=torch.float16, device=t.hi.device)
    check(_lib.lib().amdnuwa_hilo_to_f16(_p(t.hi), _p(t.lo), _ld(t.hi), _p(out), _ld(out), R, Cc, _stream()), 'amdnuwa_hilo_to_f16')
    return out


def gemm_nt(A, B, *, out=None, out_bf16=False, bias=None, alpha=1.0, shift=None, N=None, K=None, geglu_out=None, out_f16=False):
    """C[M,N] = alpha * A[M,K] @ B[N,K]^T (+ bias).  A, B: BF pairs of 2-D views.
    out: fp32 tensor view or BF pair view (allocated when None).  shift = (ntok, fmap) folds the
    token shift into A's loader.  out_f16 (hi + lo operands, bf16 output): the result is BF(hi, None, f16) -- a bf16 copy and
    an fp16 copy of the product, the operand form of the fp16 attention cores."""
    if out_f16:
        assert out is None and out_bf16 and geglu_out is None and A.lo is not None and B.lo is not None
        return _gemm_nt_f16(A, B, bias=bias, alpha=alpha, shift=shift, N=N, K=K)
    L = _lib.lib()
    M = A.hi.shape[0]
    K = A.hi.shape[1] if K is None else K
    N = B.hi.shape[0] if N is None else N
    dev = A.hi.device
    _chk_dev(A.hi, B.hi)
    x3 = A.lo is not None and B.lo is not None
    if out is None:
        out = empty_bf((M, N), dev, lo=x3 or want_lo()) if out_bf16 else torch.empty((M, N), dtype=torch.float32, device=dev)
    d = GemmDesc()
    d.A, d.Alo, d.lda = _p(A.hi), _p(A.lo) if x3 else None, _ld(A.hi)
    d.B, d.Blo, d.ldb = _p(B.hi), _p(B.lo) if x3 else None, _ld(B.hi)
    if out_bf16:
        d.C, d.Clo, d.ldc, d.c_is_bf16 = _p(out.hi), _p(out.lo), _ld(out.hi), 1
    else:
        d.C, d.Clo, d.ldc, d.c_is_bf16 = _p(out), None, _ld(out), 0
    d.bias = _p(bias)
    d.alpha, d.beta = float(alpha), 0.0
    d.M, d.N, d.K, d.batch = M, N, K, 1
    if shift is not None:
        d.shift_ntok, d.shift_fmap = int(shift[0]), int(shift[1])
    if geglu_out is not None:      # C (bf16, interleaved-by-8 columns) = u; geglu_out BF [M, N/2] = a * gelu(gate), in the epilogue when possible
        assert out_bf16
        d.C2, d.C2lo, d.ldc2 = _p(geglu_out.hi), _p(geglu_out.lo), _ld(geglu_out.hi)
        if x3 and mixed() and out.lo is not None and L.amdnuwa_gemm_nt_fused(C.byref(d)):
            # 'bf16x3-fwd': the gate runs on the fp32 accumulators inside the epilogue and the bf16 backward reads u.hi only --
            # u's lo part (a third of this GEMM's output bytes) is never needed, so it is not written
            out = BF(out.hi, None)
            d.Clo = None
    st = _stream()
    if _TIMER['on']:
        _TIMER['flops'] += 2.0 * M * N * K                     # ALGORITHMIC: one product per (m, n, k) whatever the operand form
        _TIMER['issued'] = _TIMER.get('issued', 0.) + 2.0 * M * N * K * (3 if x3 else 1)
        ob = (2 * (2 if out.lo is not None else 1)) if out_bf16 else 4
        _TIMER['bytes'] += (2.0 * (M + N) * K) * (2 if x3 else 1) + float(M) * N * ob + (float(M) * N if geglu_out is not None else 0.)
        _TIMER['tags'].append('gemm_nt'); L.amdnuwa_timer_begin(st)
    check(L.amdnuwa_gemm_nt(C.byref(d), st), 'amdnuwa_gemm_nt')
    if _TIMER['on']:
        L.amdnuwa_timer_end(st)
    return out


def _gemm_nt_f16(A, B, *, bias, alpha, shift, N, K):
    L = _lib.lib()
    M = A.hi.shape[0]
    K = A.hi.shape[1] if K is None else K
    N = B.hi.shape[0] if N is None else N
    dev = A.hi.device
    hi = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    f16 = torch.empty((M, N), dtype=torch.float16, device=dev)
    d = GemmDesc()
    d.A, d.Alo, d.lda = _p(A.hi), _p(A.lo), _ld(A.hi)
    d.B, d.Blo, d.ldb = _p(B.hi), _p(B.lo), _ld(B.hi)
    d.C, d.Clo, d.ldc, d.c_is_bf16, d.c_lo_f16 = _p(hi), _p(f16), N, 1, 1
    d.bias, d.alpha, d.beta = _p(bias), float(alpha), 0.0
    d.M, d.N, d.K, d.batch = M, N, K, 1
    if shift is not None:
        d.shift_ntok, d.shift_fmap = int(shift[0]), int(shift[1])
    if not L.amdnuwa_gemm_nt_f16_fused(C.byref(d)):
        # shapes outside the hi + lo ring (small M, token shift in the loader): plain hi + lo product, then one conversion pass
        full = gemm_nt(A, B, out_bf16=True, bias=bias, alpha=alpha, shift=shift, N=N, K=K)
        return BF(full.hi, None, hilo_to_f16(full))
    st = _stream()
    if _TIMER['on']:
        _TIMER['flops'] += 2.0 * M * N * K
        _TIMER['issued'] = _TIMER.get('issued', 0.) + 6.0 * M * N * K
        _TIMER['bytes'] += 4.0 * (M + N) * K + 4.0 * M * N
        _TIMER['tags'].append('gemm_nt'); L.amdnuwa_timer_begin(st)
    check(L.amdnuwa_gemm_nt(C.byref(d), st), 'amdnuwa_gemm_nt(f16 copy)')
    if _TIMER['on']:
        L.amdnuwa_timer_end(st)
    return BF(hi, None, f16)


_FF_F16 = os.environ.get('AMDNUWA_F16_FF', '1') != '0'


def set_ff_f16(on):
    """'bf16x3-fwd' only: run the FeedForward GEMMs of the forward on single fp16 MFMAs (default) or, when off, as 3-MFMA bf16 hi + lo
    products like the other projections (A/B switch; the full-depth logits stay within 1e-3 either way)"""
    global _FF_F16
    _FF_F16 = bool(on)


def ff_f16():
    return _FF_F16 and mixed()


def _f16ops_desc(A16, B16, M, N, Kd):
    d = GemmDesc()
    d.A, d.lda, d.B, d.ldb = _p(A16), _ld(A16), _p(B16), _ld(B16)
    d.alpha, d.beta = 1.0, 0.0
    d.M, d.N, d.K, d.batch, d.ab_f16 = M, N, Kd, 1, 1
    return d


def gemm_nt_f16ops_ok(M, N, Kd, *, out_bf16, gate=False, out_f16=False, geglu_bwd=False):
    """will amdnuwa_gemm_nt take this product with fp16 operands (the 256x256 ring at training sizes)?  out_f16: one fp16 output;
    geglu_bwd: the product is dgg [M, N] and the epilogue writes du fp16 [M, 2 N] from u [M, 2 N]"""
    d = GemmDesc()
    one = ctypes_dummy()
    d.A, d.B, d.C = one, one, one
    d.lda, d.ldb, d.ldc = Kd, Kd, N
    d.c_is_bf16 = 1 if (out_bf16 or out_f16 or geglu_bwd) else 0
    d.c_f16 = 1 if (out_f16 or geglu_bwd) else 0
    if geglu_bwd:
        d.C2, d.ldc2, d.geglu_u, d.ld_u = one, 2 * N, one, 2 * N
    if gate:
        d.C2, d.ldc2 = one, N // 2
    d.M, d.N, d.K, d.batch, d.ab_f16 = M, N, Kd, 1, 1
    return bool(_lib.lib().amdnuwa_gemm_nt_f16ops_supported(C.byref(d)))


def ctypes_dummy():
    return 16          # any non-null, 16-byte aligned address: the query never dereferences


_QKV_F16 = os.environ.get('AMDNUWA_F16_QKV', '1') != '0'


def set_qkv_f16(on):
    """'bf16x3-fwd' only: the Sparse3DNA q / k / v projection of the forward on single fp16 MFMAs (default) or as a 3-MFMA product"""
    global _QKV_F16
    _QKV_F16 = bool(on)


def qkv_f16():
    return _QKV_F16 and cores_f16()


def _x2_classes(v):
    if v in (True, '1', 'all'):
        return frozenset('oql')
    if v in (False, None, '0', ''):
        return frozenset()
    assert set(v) <= set('oql'), v
    return frozenset(v)


# default: to_out x2 and the cross-attention q projection.  Measured on the full-depth logits (tests/test_gpu_named_size.py, rel-l2 against the
# fp32 oracle): none 6.2e-4, 'o' 6.6e-4, 'oq' 6.7e-4, 'oql' 7.0e-4 -- to_logits buys 1.5 ms of the 10 for a third of the added error and stays
# a three-MFMA product.
DEFAULT_F16X2 = 'oq'
_PROJ_F16X2 = _x2_classes(os.environ.get('AMDNUWA_F16X2', DEFAULT_F16X2))


def set_proj_f16x2(on):
    """'bf16x3-fwd' only: the products that are still hi + lo pairs on both sides as TWO fp16 MFMAs -- the activation as ONE fp16 value,
    the weight as an fp16 hi + lo pair (exact to ~22 bits) -- instead of three bf16 MFMAs.  Priced by tools/error_budget.py before it
    was built.  on: True / False, or a string of classes: 'o' = to_out of both attention blocks, 'q' = the
    cross-attention q projection, 'l' = to_logits (env AMDNUWA_F16X2, same values; '0' = off; default DEFAULT_F16X2)."""
    global _PROJ_F16X2
    _PROJ_F16X2 = _x2_classes(on)


def proj_f16x2(cls=None):
    """is the two-MFMA form on (for product class `cls`; None: for any)?"""
    return mixed() and (bool(_PROJ_F16X2) if cls is None else cls in _PROJ_F16X2)


def f16_pair(w):
    """fp32 weight -> (hi, lo) fp16 tensors with hi + lo = w to ~22 significand bits (lo may be subnormal: the fp16 MFMA keeps them)"""
    w = w.detach().float()
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)
    return hi.contiguous(), lo.contiguous()


def gemm_nt_f16x2_ok(M, N, Kd, *, out_bf16):
    d = GemmDesc()
    one = ctypes_dummy()
    d.A, d.B, d.Blo, d.C = one, one, one, one
    d.lda, d.ldb, d.ldc = Kd, Kd, N
    d.c_is_bf16 = 1 if out_bf16 else 0
    d.M, d.N, d.K, d.batch, d.ab_f16 = M, N, Kd, 1, 1
    return bool(_lib.lib().amdnuwa_gemm_nt_f16x2_supported(C.byref(d)))


def gemm_nt_f16x2(A16, B16, *, out_bf16=False, bias=None, copy_f16=False, out_f16=False):
    """two-MFMA product: A16 fp16 [M, K] x (B16 = (hi, lo) fp16 [N, K] pair)^T.  out_bf16=False -> fp32 [M, N] (+ bias);
    out_bf16=True -> BF(bf16 copy, None, fp16 copy if copy_f16); out_f16=True -> ONE fp16 [M, N] tensor (saturating)"""
    L = _lib.lib()
    M, Kd = A16.shape
    Bh, Bl = B16
    N = Bh.shape[0]
    dev = A16.device
    d = GemmDesc()
    d.A, d.lda, d.B, d.Blo, d.ldb = _p(A16), _ld(A16), _p(Bh), _p(Bl), _ld(Bh)
    d.alpha, d.beta, d.bias = 1.0, 0.0, _p(bias)
    d.M, d.N, d.K, d.batch, d.ab_f16 = M, N, Kd, 1, 1
    c16 = None
    if out_f16:
        assert not (out_bf16 or copy_f16)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        d.C, d.ldc, d.c_is_bf16, d.c_f16 = _p(out), N, 1, 1
        out_bf16 = True
    elif out_bf16:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        d.C, d.ldc, d.c_is_bf16 = _p(out), N, 1
        if copy_f16:
            c16 = torch.empty((M, N), dtype=torch.float16, device=dev)
            d.Clo = _p(c16)
    else:
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
        d.C, d.ldc, d.c_is_bf16 = _p(out), N, 0
    st = _stream()
    if _TIMER['on']:
        _TIMER['flops'] += 2.0 * M * N * Kd
        _TIMER['issued'] = _TIMER.get('issued', 0.) + 4.0 * M * N * Kd
        _TIMER['bytes'] += 2.0 * M * Kd + 4.0 * N * Kd + float(M) * N * ((4 if c16 is not None else 2) if out_bf16 else 4)
        _TIMER['tags'].append('gemm_nt'); L.amdnuwa_timer_begin(st)
    check(L.amdnuwa_gemm_nt(C.byref(d), st), 'amdnuwa_gemm_nt(f16 x f16 hi+lo)')
    if _TIMER['on']:
        L.amdnuwa_timer_end(st)
    if out_f16:
        return out
    return BF(out, None, c16) if out_bf16 else out


def gemm_nt_f16ops(A16, B16, *, out_bf16=False, gate=False, copy_f16=False, gate_bf16=True, out_f16=False):
    """fp16-operand product on the fp16 MFMA.  A16 [M, K], B16 [N, K] fp16 tensors.
    out_bf16=False -> fp32 [M, N].  out_bf16=True -> u bf16 [M, N]; with gate=True also (gg16 fp16 [M, N/2], gg bf16 [M, N/2] or None when
    gate_bf16 is off) = a * gelu(gate) computed on the fp32 accumulators (u in the interleaved-by-8 layout); with copy_f16=True ->
    BF(bf16 copy, None, fp16 copy).  out_f16=True -> ONE fp16 [M, N] output, saturating (dgrad products of the fp16-gradient backward)."""
    L = _lib.lib()
    M, Kd = A16.shape
    N = B16.shape[0]
    dev = A16.device
    d = _f16ops_desc(A16, B16, M, N, Kd)
    gg16 = ggb = c16 = None
    if out_f16:
        assert not (out_bf16 or gate or copy_f16)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        d.C, d.ldc, d.c_is_bf16, d.c_f16 = _p(out), N, 1, 1
    elif out_bf16:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        d.C, d.ldc, d.c_is_bf16 = _p(out), N, 1
        if copy_f16:
            assert not gate
            c16 = torch.empty((M, N), dtype=torch.float16, device=dev)
            d.Clo = _p(c16)
        if gate:
            gg16 = torch.empty((M, N // 2), dtype=torch.float16, device=dev)
            ggb = torch.empty((M, N // 2), dtype=torch.bfloat16, device=dev) if gate_bf16 else None
            d.C2, d.C2lo, d.ldc2 = _p(gg16), _p(ggb), N // 2
    else:
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
        d.C, d.ldc, d.c_is_bf16 = _p(out), N, 0
    st = _stream()
    if _TIMER['on']:
        _TIMER['flops'] += 2.0 * M * N * Kd
        _TIMER['issued'] = _TIMER.get('issued', 0.) + 2.0 * M * N * Kd
        _TIMER['bytes'] += 2.0 * (M + N) * Kd + float(M) * N * (2 if (out_bf16 or out_f16) else 4) + ((2.0 if gate_bf16 else 1.0) * M * N if gate else 0.)
        _TIMER['tags'].append('gemm_nt'); L.amdnuwa_timer_begin(st)
    check(L.amdnuwa_gemm_nt(C.byref(d), st), 'amdnuwa_gemm_nt(fp16 operands)')
    if _TIMER['on']:
        L.amdnuwa_timer_end(st)
    if copy_f16:
        return BF(out, None, c16)
    return (out, gg16, ggb) if gate else out


def gemm_nt_geglu_bwd(dy, w2T, u, FP):
    """FF backward through the gate: dgg = dy @ w2T^T [M, FP] and du = (dgg * gelu(gate) | dgg * a * gelu'(gate)) in u's interleaved
    layout.  On the 256x256 ring the gate runs in the GEMM epilogue and dgg never exists in memory; otherwise GEMM + gate kernel."""
    L = _lib.lib()
    M, Kd = dy.hi.shape
    dev = dy.hi.device
    x3 = dy.lo is not None and w2T.lo is not None
    du = empty_bf((M, 2 * FP), dev, lo=u.lo is not None)
    d = GemmDesc()
    d.A, d.Alo, d.lda = _p(dy.hi), _p(dy.lo) if x3 else None, _ld(dy.hi)
    d.B, d.Blo, d.ldb = _p(w2T.hi), _p(w2T.lo) if x3 else None, _ld(w2T.hi)
    d.c_is_bf16, d.ldc = 1, FP
    d.alpha, d.beta = 1.0, 0.0
    d.M, d.N, d.K, d.batch = M, FP, Kd, 1
    d.C2, d.C2lo, d.ldc2 = _p(du.hi), _p(du.lo), 2 * FP
    d.geglu_u, d.geglu_u_lo, d.ld_u = _p(u.hi), _p(u.lo), _ld(u.hi)
    d.C = _p(du.hi)                                   # placeholder: not written when the epilogue is fused
    if not L.amdnuwa_gemm_nt_fused(C.byref(d)):
        dgg = empty_bf((M, FP), dev, lo=x3 or want_lo())
        d.C, d.Clo = _p(dgg.hi), _p(dgg.lo)
    st = _stream()
    if _TIMER['on']:
        _TIMER['flops'] += 2.0 * M * FP * Kd
        _TIMER['issued'] = _TIMER.get('issued', 0.) + 2.0 * M * FP * Kd * (3 if x3 else 1)
        _TIMER['bytes'] += (2.0 * (M + FP) * Kd) * (2 if x3 else 1) + float(M) * FP * 8
        _TIMER['tags'].append('gemm_nt'); L.amdnuwa_timer_begin(st)
    check(L.amdnuwa_gemm_nt(C.byref(d), st), 'amdnuwa_gemm_nt(geglu backward)')
    if _TIMER['on']:
        L.amdnuwa_timer_end(st)
    return du


def gemm_nt_geglu_bwd16(dy16, w2T16, u, FP):
    """fp16-gradient form of gemm_nt_geglu_bwd: dy16 fp16 [M, D] (= S * dy), w2T16 fp16 [FP, D], u bf16 [M, 2 FP] (interleaved) ->
    du fp16 [M, 2 FP] (= S * du), the gate's backward in the GEMM epilogue"""
    L = _lib.lib()
    M, Kd = dy16.shape
    du = torch.empty((M, 2 * FP), dtype=torch.float16, device=dy16.device)
    d = _f16ops_desc(dy16, w2T16, M, FP, Kd)
    d.c_is_bf16, d.c_f16, d.ldc = 1, 1, FP
    d.C2, d.ldc2 = _p(du), 2 * FP
    d.geglu_u, d.ld_u = _p(u), _ld(u)
    d.C = _p(du)                                      # placeholder: the fused epilogue does not write C
    st = _stream()
    if _TIMER['on']:
        _TIMER['flops'] += 2.0 * M * FP * Kd
        _TIMER['issued'] = _TIMER.get('issued', 0.) + 2.0 * M * FP * Kd
        _TIMER['bytes'] += 2.0 * (M + FP) * Kd + float(M) * FP * 8
        _TIMER['tags'].append('gemm_nt'); L.amdnuwa_timer_begin(st)
    check(L.amdnuwa_gemm_nt(C.byref(d), st), 'amdnuwa_gemm_nt(geglu backward, fp16)')
    if _TIMER['on']:
        L.amdnuwa_timer_end(st)
    return du


def _tn16_desc(A16, B16, out, N1, N2):
    d = GemmDesc()
    d.A, d.lda, d.B, d.ldb = _p(A16), _ld(A16), _p(B16), _ld(B16)
    d.C, d.ldc, d.c_is_bf16 = _p(out) if out is not None else ctypes_dummy(), (_ld(out) if out is not None else N2), 0
    d.M, d.N, d.K, d.batch, d.ab_f16 = N1, N2, A16.shape[0], 1, 1
    return d


def gemm_tn16_ok(R, N1, N2, lda=None, ldb=None):
    """does amdnuwa_gemm_tn take [R, N1]^T [R, N2] with fp16 operands (the four-wave kernel's shapes)?"""
    d = GemmDesc()
    one = ctypes_dummy()
    d.A, d.B, d.C = one, one, one
    d.lda, d.ldb, d.ldc = (N1 if lda is None else lda), (N2 if ldb is None else ldb), N2
    d.M, d.N, d.K, d.batch, d.ab_f16 = N1, N2, R, 1, 1
    return bool(_lib.lib().amdnuwa_gemm_tn_f16_supported(C.byref(d)))


def gemm_tn16(A16, B16, out, s2, *, alpha=1.0, beta=0.0, N1=None, N2=None):
    """out[N1, N2] (fp32) = beta * out + alpha / S * A16[R, N1]^T @ B16[R, N2]: a weight gradient from an fp16 gradient (A16 = S * dY) and
    an fp16 activation copy; s2 = device {S, 1 / S} (None: no scale)"""
    L = _lib.lib()
    N1 = A16.shape[1] if N1 is None else N1
    N2 = B16.shape[1] if N2 is None else N2
    d = _tn16_desc(A16, B16, out, N1, N2)
    d.alpha, d.beta = float(alpha), float(beta)
    d.alpha_dev = None if s2 is None else s2.data_ptr() + 4
    nb = L.amdnuwa_gemm_tn_workspace_bytes(C.byref(d))
    ws = workspace(nb, A16.device)
    check(L.amdnuwa_gemm_tn(C.byref(d), _p(ws), nb, _stream()), 'amdnuwa_gemm_tn(fp16)')
    return out


def gemm_tn(A, B, out, *, alpha=1.0, beta=0.0, shift=None, N1=None, N2=None):
    """out[N1,N2] (fp32 view) = beta*out + alpha * A[R,N1]^T @ B[R,N2]; shift applies to B's loader."""
    L = _lib.lib()
    R = A.hi.shape[0]
    N1 = A.hi.shape[1] if N1 is None else N1
    N2 = B.hi.shape[1] if N2 is None else N2
    x3 = A.lo is not None and B.lo is not None
    d = GemmDesc()
    d.A, d.Alo, d.lda = _p(A.hi), _p(A.lo) if x3 else None, _ld(A.hi)
    d.B, d.Blo, d.ldb = _p(B.hi), _p(B.lo) if x3 else None, _ld(B.hi)
    d.C, d.ldc, d.c_is_bf16 = _p(out), _ld(out), 0
    d.alpha, d.beta = float(alpha), float(beta)
    d.M, d.N, d.K, d.batch = N1, N2, R, 1
    if shift is not None:
        d.shift_ntok, d.shift_fmap = int(shift[0]), int(shift[1])
    nb = L.amdnuwa_gemm_tn_workspace_bytes(C.byref(d))
    ws = workspace(nb, A.hi.device)
    check(L.amdnuwa_gemm_tn(C.byref(d), _p(ws), nb, _stream()), 'amdnuwa_gemm_tn')
    return out


def gemm_tn_batched(desc, device):
    L = _lib.lib()
    nb = L.amdnuwa_gemm_tn_workspace_bytes(C.byref(desc))
    ws = workspace(nb, device)
    check(L.amdnuwa_gemm_tn(C.byref(desc), _p(ws), nb, _stream()), 'amdnuwa_gemm_tn(batched)')


# ------------------------------------------------------------------------------------------------
# row kernels
# ------------------------------------------------------------------------------------------------

LN_X_BF16, LN_DY_BF16, LN_LO_F16 = 16, 32, 64       # == AMDNUWA_LN_X_BF16 / AMDNUWA_LN_DY_BF16 / AMDNUWA_LN_LO_F16
LN_OUT_F16, LN_DY_F16, LN_DY_SCALED = 128, 256, 512   # == AMDNUWA_LN_OUT_F16 / AMDNUWA_LN_DY_F16 / AMDNUWA_LN_DY_SCALED
LN_RESID_MINUS = 1024                                # == AMDNUWA_LN_RESID_MINUS


def _f32_or_bf(t):
    """(data pointer, is_bf16, shape, device) of an fp32 tensor or a BF pair without lo part"""
    if isinstance(t, BF):
        assert t.lo is None, 'bf16 inputs to the LN kernels are the fast-mode (hi only) form'
        return _p(t.hi), True, t.hi.shape, t.hi.device
    return _p(t), False, t.shape, t.device


def empty_bf_f16(shape, device):
    """BF(hi = bf16 copy, None, f16 = fp16 copy)"""
    return BF(torch.empty(shape, dtype=torch.bfloat16, device=device), None, torch.empty(shape, dtype=torch.float16, device=device))


@_family('ln')
def ln_fwd(x, w, b, *, resid=None, stable=False, eps=1e-5, shift=None, f16=False, minus=False):
    """x fp32 [R, D] contiguous (or a hi-only BF pair).  resid None -> (BF out, mean, rstd, inv_amax);
    else (fp32 out = resid + LN(x) -- minus: resid - LN(x) --, mean, rstd).  f16: out = BF(hi, None, f16) (bf16 copy + fp16 copy)"""
    L = _lib.lib()
    xp, xbf, (R, D), dev = _f32_or_bf(x)
    flag = LN_X_BF16 if xbf else 0
    mean = torch.empty(R, dtype=torch.float32, device=dev)
    rstd = torch.empty(R, dtype=torch.float32, device=dev)
    if resid is None:
        ia = torch.empty(R, dtype=torch.float32, device=dev) if stable else None
        sn, sf = (int(shift[0]), int(shift[1])) if shift is not None else (0, 0)      # out = shift(LN(x)) (ShiftVideoTokens)
        if f16 == 'only':                        # ONE fp16 copy (the block's backward runs on fp16 operands too)
            out = BF(None, None, torch.empty((R, D), dtype=torch.float16, device=dev))
            first, second, fl = out.f16, None, LN_OUT_F16
        else:
            out = empty_bf_f16((R, D), dev) if f16 else empty_bf((R, D), dev)
            first, second, fl = out.hi, (out.f16 if f16 else out.lo), (LN_LO_F16 if f16 else 0)
        check(L.amdnuwa_ln_fwd(xp, None, _p(w), _p(b), _p(first), _p(second), None, _p(mean), _p(rstd), _p(ia),
                               R, D, 0 | flag | fl, 1 if stable else 0, eps, sn, sf, _stream()), 'amdnuwa_ln_fwd')
        return out, mean, rstd, ia
    out = torch.empty_like(resid)
    check(L.amdnuwa_ln_fwd(xp, _p(resid), _p(w), _p(b), None, None, _p(out), _p(mean), _p(rstd), None,
                           R, D, 1 | flag | (LN_RESID_MINUS if minus else 0), 0, eps, 0, 0, _stream()), 'amdnuwa_ln_fwd')
    return out, mean, rstd


@_family('ln')
def ln_post_pre_fwd(y, resid, w, b, next_w, next_b, *, eps=1e-5, next_shift=None, next_f16=False):
    """post-norm + residual of one block and the pre-norm (+ token shift) of the next in one pass over the stream:
    returns (out fp32 = resid + LN(y; w, b), mean, rstd, h BF = shift(LN(out; next_w, next_b)), next_mean, next_rstd)"""
    L = _lib.lib()
    yp, ybf, (R, D), dev = _f32_or_bf(y)
    mean, rstd, mean2, rstd2 = (torch.empty(R, dtype=torch.float32, device=dev) for _ in range(4))
    out = torch.empty_like(resid)
    sn, sf = (int(next_shift[0]), int(next_shift[1])) if next_shift is not None else (0, 0)
    if next_f16 == 'only':
        h = BF(None, None, torch.empty((R, D), dtype=torch.float16, device=dev))
        first, second, fl = h.f16, None, LN_OUT_F16
    else:
        h = empty_bf_f16((R, D), dev) if next_f16 else empty_bf((R, D), dev)
        first, second, fl = h.hi, (h.f16 if next_f16 else h.lo), (LN_LO_F16 if next_f16 else 0)
    check(L.amdnuwa_ln_post_pre_fwd(yp, _p(resid), _p(w), _p(b), _p(out), _p(mean), _p(rstd), _p(next_w), _p(next_b),
                                    _p(first), _p(second), _p(mean2), _p(rstd2), R, D,
                                    (LN_X_BF16 if ybf else 0) | fl, eps, sn, sf,
                                    _stream()), 'amdnuwa_ln_post_pre_fwd')
    return out, mean, rstd, h, mean2, rstd2


@_family('ln')
def ln_bwd(dy, x, mean, rstd, w, *, inv_amax=None, to_bf=False, dres=None, shift=None, want_dsum=False, to_f16=None, dy_scale2=None):
    """returns (dx, dw, db, dsum).  to_bf: dx as BF pair; to_f16 = s2 (device {S, 1 / S}): dx as G16 = fp16(S * dx); else dx fp32 =
    dres + dx_ln (dres may be None -> zeros).  dy / x: fp32 tensors, or (one of them) a hi-only BF pair; dy may be a G16."""
    L = _lib.lib()
    s2 = to_f16
    if isinstance(dy, G16):
        dyp, dybf, s2 = _p(dy.t), False, dy.s2
    else:
        dyp, dybf, _, _ = _f32_or_bf(dy)
        if dy_scale2 is not None:                # fp32 dy multiplied by the DEVICE scalar dy_scale2[1] on the way in
            assert not dybf and to_f16 is None
            s2 = dy_scale2
    xp, xbf, (R, D), dev = _f32_or_bf(x)
    st = (1 if inv_amax is not None else 0) | (LN_X_BF16 if xbf else 0) | (LN_DY_BF16 if dybf else 0) | (LN_DY_F16 if isinstance(dy, G16) else 0) | \
        (LN_DY_SCALED if dy_scale2 is not None else 0)
    dw = torch.empty(D, dtype=torch.float32, device=dev)
    db = torch.empty(D, dtype=torch.float32, device=dev)
    ds = torch.empty(D, dtype=torch.float32, device=dev) if want_dsum else None
    nb = L.amdnuwa_ln_bwd_workspace_bytes(R, D)
    ws = workspace(nb, dev)
    sn, sf = (int(shift[0]), int(shift[1])) if shift is not None else (0, 0)
    if to_f16 is not None:
        dx = G16(torch.empty((R, D), dtype=torch.float16, device=dev), to_f16)
        check(L.amdnuwa_ln_bwd_f16(dyp, xp, _p(mean), _p(rstd), _p(inv_amax), _p(w), _p(dx.t), None, None, None,
                                   _p(dw), _p(db), _p(ds), R, D, sn, sf, st | LN_OUT_F16, 0, _p(s2), _p(ws), nb, _stream()), 'amdnuwa_ln_bwd_f16')
    elif to_bf:
        dx = empty_bf((R, D), dev)
        check(L.amdnuwa_ln_bwd_f16(dyp, xp, _p(mean), _p(rstd), _p(inv_amax), _p(w), _p(dx.hi), _p(dx.lo), None, None,
                                   _p(dw), _p(db), _p(ds), R, D, sn, sf, st, 0, _p(s2), _p(ws), nb, _stream()), 'amdnuwa_ln_bwd')
    else:
        if dres is None:
            dx = torch.zeros((R, D), dtype=torch.float32, device=dev)
            dr = None
        else:
            dx = torch.empty((R, D), dtype=torch.float32, device=dev)
            dr = dres
        check(L.amdnuwa_ln_bwd_f16(dyp, xp, _p(mean), _p(rstd), _p(inv_amax), _p(w), None, None, _p(dx), _p(dr),
                                   _p(dw), _p(db), _p(ds), R, D, sn, sf, st, 0, _p(s2), _p(ws), nb, _stream()), 'amdnuwa_ln_bwd')
    return dx, dw, db, ds


@_family('ln')
def ln_bwd_chain(dh, x, mean, rstd, w, g, y_prev, mean_prev, rstd_prev, w_prev, *, shift=None, want_dsum=False, out_f16=None):
    """pre-norm backward of a block and the post-norm backward of the block before it in one pass over the gradient row.
    dh, y_prev: both fp32 tensors, both hi-only BF pairs, or dh a BF pair / a G16 with an fp32 y_prev ('bf16x3-fwd').  out_f16 = s2: dy_prev
    leaves as a G16.  returns (dx fp32, dw, db, dy_prev BF | G16, dw_prev, db_prev, dsum_prev)"""
    L = _lib.lib()
    s2 = out_f16
    yp, ybf, (R, D), dev = _f32_or_bf(y_prev)
    if isinstance(dh, G16):
        assert not ybf
        dhp, form, s2 = _p(dh.t), 3, dh.s2
        assert out_f16 is None or out_f16.data_ptr() == dh.s2.data_ptr(), 'one gradient scale per backward pass'
    else:
        dhp, dhbf, _, _ = _f32_or_bf(dh)
        assert dhbf or not ybf, 'ln_bwd_chain: fp32 dh with a bf16 y_prev is not a form any mode produces'
        form = 1 if (dhbf and ybf) else (2 if dhbf else 0)            # == the inputs_bf16 argument of amdnuwa_ln_bwd_chain
    dx = torch.empty((R, D), dtype=torch.float32, device=dev)
    dw, db, dwp, dbp = (torch.empty(D, dtype=torch.float32, device=dev) for _ in range(4))
    dsp = torch.empty(D, dtype=torch.float32, device=dev) if want_dsum else None
    if out_f16 is not None:
        dyp = G16(torch.empty((R, D), dtype=torch.float16, device=dev), out_f16)
        o_hi, o_lo = dyp.t, None
    else:
        dyp = empty_bf((R, D), dev)
        o_hi, o_lo = dyp.hi, dyp.lo
    nb = L.amdnuwa_ln_bwd_chain_workspace_bytes(R, D)
    ws = workspace(nb, dev)
    sn, sf = (int(shift[0]), int(shift[1])) if shift is not None else (0, 0)
    check(L.amdnuwa_ln_bwd_chain_f16(dhp, _p(x), _p(mean), _p(rstd), _p(w), _p(g), _p(dx), _p(dw), _p(db), yp, _p(mean_prev),
                                     _p(rstd_prev), _p(w_prev), _p(o_hi), _p(o_lo), _p(dwp), _p(dbp), _p(dsp), R, D, sn, sf,
                                     form, 1 if out_f16 is not None else 0, _p(s2), _p(ws), nb, _stream()), 'amdnuwa_ln_bwd_chain')
    return dx, dw, db, dyp, dwp, dbp, dsp


def colsum(x):
    L = _lib.lib()
    R, D = x.shape
    out = torch.empty(D, dtype=torch.float32, device=x.device)
    nb = L.amdnuwa_colsum_workspace_bytes(R, D)
    ws = workspace(nb, x.device)
    check(L.amdnuwa_colsum(_p(x), _p(out), R, D, 0, _p(ws), nb, _stream()), 'amdnuwa_colsum')
    return out


def geglu_fwd(u, FP, interleaved=False):
    """u BF [R, 2*FP] -> a * gelu(gate) BF [R, FP].  interleaved: u in the interleaved-by-8 layout (see geglu_interleave)"""
    L = _lib.lib()
    R = u.hi.shape[0]
    out = empty_bf((R, FP), u.hi.device, lo=u.lo is not None)
    fn = L.amdnuwa_geglu_il_fwd if interleaved else L.amdnuwa_geglu_fwd
    check(fn(_p(u.hi), _p(u.lo), _p(out.hi), _p(out.lo), R, FP, _stream()), 'amdnuwa_geglu_fwd')
    return out


def geglu_bwd(u, dgg, FP, interleaved=False):
    L = _lib.lib()
    R = u.hi.shape[0]
    du = empty_bf((R, 2 * FP), u.hi.device, lo=u.lo is not None)
    fn = L.amdnuwa_geglu_il_bwd if interleaved else L.amdnuwa_geglu_bwd
    check(fn(_p(u.hi), _p(u.lo), _p(dgg.hi), _p(dgg.lo), _p(du.hi), _p(du.lo), R, FP, _stream()), 'amdnuwa_geglu_bwd')
    return du


def geglu_interleave(t, FP, dim=0):
    """[a (FP) | gate (FP)] along `dim` -> the interleaved-by-8 order: 8 values, their 8 gates, the next 8 values, ...
    (FP % 8 == 0).  geglu_deinterleave is the inverse."""
    shp = list(t.shape)
    assert shp[dim] == 2 * FP and FP % 8 == 0
    v = t.reshape(shp[:dim] + [2, FP // 8, 8] + shp[dim + 1:])
    return v.transpose(dim, dim + 1).reshape(shp)


def geglu_deinterleave(t, FP, dim=0):
    shp = list(t.shape)
    assert shp[dim] == 2 * FP and FP % 8 == 0
    v = t.reshape(shp[:dim] + [FP // 8, 2, 8] + shp[dim + 1:])
    return v.transpose(dim, dim + 1).reshape(shp)


def cast_pad(src, out, row0=0, Cp=None):
    """out[row0 + r, :Cp] = bf16(src[r, :]) (zero padded); src fp32 2-D view, out BF pair (2-D)"""
    L = _lib.lib()
    R, Cc = src.shape
    Cp = out.hi.shape[1] if Cp is None else Cp
    hi = out.hi[row0:row0 + R]
    lo = None if out.lo is None else out.lo[row0:row0 + R]
    check(L.amdnuwa_cast_pad(_p(src), src.stride(0), _p(hi), _p(lo), out.hi.stride(0), R, Cc, Cp, _stream()), 'amdnuwa_cast_pad')


def transpose_cast(src, out, col0=0):
    """out[c, col0 + r] = bf16(src[r, c])"""
    L = _lib.lib()
    R, Cc = src.shape
    hi = out.hi[:, col0:]
    lo = None if out.lo is None else out.lo[:, col0:]
    check(L.amdnuwa_transpose_cast(_p(src), src.stride(0), _p(hi), _p(lo), out.hi.stride(0), R, Cc, _stream()),
          'amdnuwa_transpose_cast')


def embed_fwd(ids, W, ax1, ax2, ax3, bos, B, ntok, H, Wd, frac):
    L = _lib.lib()
    D = W.shape[1]
    x = torch.empty((B * ntok, D), dtype=torch.float32, device=W.device)
    check(L.amdnuwa_embed_fwd(_p(ids), _p(W), _p(ax1), _p(ax2), _p(ax3), _p(bos), _p(x), B, ntok, D, H, Wd, float(frac),
                              _stream()), 'amdnuwa_embed_fwd')
    return x


def embed_bwd(ids, dx, dW, dax1, dax2, dax3, dbos, B, ntok, F, H, Wd, frac):
    L = _lib.lib()
    D = dW.shape[1]
    nb = L.amdnuwa_embed_bwd_workspace_bytes(ntok, D)
    # (the token-gradient kernel cuts the sorted ids into as many segments as the workspace holds two partial rows for: room for 64-row
    #  segments keeps a wave's serial walk short; the minimum the library asks for still works, with longer segments)
    nb = max(nb, (2 * ((B * (ntok - 1) + 63) // 64) + 64) * (D + 2) * 4)
    ws = workspace(nb, dx.device)
    sid, perm = torch.sort(ids.reshape(-1), stable=True)      # fixed summation order per embedding row: no atomics
    check(L.amdnuwa_embed_bwd(_p(ids), _p(sid), _p(perm), _p(dx), _p(dW), _p(dax1), _p(dax2), _p(dax3), _p(dbos), B, ntok, D,
                              F, H, Wd, float(frac), _p(ws), nb, _stream()), 'amdnuwa_embed_bwd')


def ce_fwd(logits, targets, grad_scale, want_grad=True, lo=None):
    L = _lib.lib()
    R, Cc = logits.shape
    dev = logits.device
    row_loss = torch.empty(R, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dl = empty_bf((R, Cc), dev, lo=lo) if want_grad else BF(None, None)
    check(L.amdnuwa_ce_fwd(_p(logits), _p(targets), _p(row_loss), _p(loss), _p(dl.hi), _p(dl.lo), R, Cc, Cc,
                           float(grad_scale), _stream()), 'amdnuwa_ce_fwd')
    return loss, dl


def linear_ce(h, w, targets, grad_scale, want_grad=True, w16=None):
    """fused to_logits + cross entropy: h [R, K], w [C, K] BF operands -> (mean loss, dlogits BF [R, C] (hi only) or BF(None, None)).
    hi-only operands run on the bf16 ring; hi + lo pairs (the logits of 'bf16x3-fwd', whose backward takes a bf16 dlogits) on the
    three-MFMA ring for the statistics / loss pass and -- when w16, an fp16 copy of w, is given -- on ONE fp16 MFMA per product for the
    dlogits pass (h's fp16 copy is made here).  The fp32 logits are never written.  Returns None when the fused kernels do not take the
    call: C % 64, K % 32, one operand with and one without a lo part, or hi + lo operands in the 'bf16x3' mode (its backward wants
    dlogits as a pair)."""
    L = _lib.lib()
    x3 = h.lo is not None and w.lo is not None
    if (h.lo is None) != (w.lo is None) or (x3 and not mixed()):
        return None
    R, Kd = h.hi.shape
    Cc = w.hi.shape[0]
    nb = L.amdnuwa_linear_ce_workspace_bytes(R, Cc)
    if nb == 0 or Kd % 32 or _ld(h.hi) % 8 or _ld(w.hi) % 8 or (x3 and (_ld(h.lo) != _ld(h.hi) or _ld(w.lo) != _ld(w.hi))):
        return None
    dev = h.hi.device
    ws = workspace(nb, dev)
    row_loss = torch.empty(R, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dl = empty_bf((R, Cc), dev, lo=False) if want_grad else BF(None, None)
    p2_f16 = x3 and want_grad and w16 is not None and _ld(w16) == _ld(w.hi)
    h16 = hilo_to_f16(h) if p2_f16 else None
    if p2_f16 and _ld(h16) != _ld(h.hi):
        p2_f16, h16 = False, None
    st = _stream()
    if _TIMER['on']:                       # two products: both count as NT GEMM work of the step
        npass = 2 if want_grad else 1
        issued = (3 if x3 else 1) + ((1 if p2_f16 else 3 if x3 else 1) if want_grad else 0)
        _TIMER['flops'] += 2.0 * R * Cc * Kd * npass
        _TIMER['issued'] = _TIMER.get('issued', 0.) + 2.0 * R * Cc * Kd * issued
        _TIMER['bytes'] += (2.0 * (R + Cc) * Kd) * issued + (2.0 * R * Cc if want_grad else 0.) + 8.0 * R * (Cc // 64)
        _TIMER['tags'].append('gemm_nt'); L.amdnuwa_timer_begin(st)
    if x3:
        check(L.amdnuwa_linear_ce_x3(_p(h.hi), _p(h.lo), _p(h16), _ld(h.hi), _p(w.hi), _p(w.lo), _p(w16) if p2_f16 else None, _ld(w.hi),
                                     _p(targets), R, Cc, Kd, float(grad_scale), _p(row_loss), _p(loss), _p(dl.hi), Cc, _p(ws), nb, st),
              'amdnuwa_linear_ce_x3')
    else:
        check(L.amdnuwa_linear_ce(_p(h.hi), _ld(h.hi), _p(w.hi), _ld(w.hi), _p(targets), R, Cc, Kd, float(grad_scale), _p(row_loss), _p(loss),
                                  _p(dl.hi), Cc, _p(ws), nb, st), 'amdnuwa_linear_ce')
    if _TIMER['on']:
        L.amdnuwa_timer_end(st)
    return loss, dl


def scale_by_device_scalar(x, scalar):
    check(_lib.lib().amdnuwa_scale_by_device_scalar(_p(x), x.numel(), _p(scalar), _stream()), 'amdnuwa_scale_by_device_scalar')


# ------------------------------------------------------------------------------------------------
# attention cores
# ------------------------------------------------------------------------------------------------

def s3_geom(B, ntok, video_shape, kernel, dilation, heads, dim_head, causal=True):
    g = S3Geom()
    g.noncausal = 0 if causal else 1
    g.B, g.ntok = B, ntok
    g.F, g.H, g.W = video_shape
    g.kf, g.kh, g.kw = kernel
    g.df, g.dh, g.dw = dilation
    g.heads, g.dim_head, g.scale = heads, dim_head, dim_head ** -0.5
    return g


def s3_supported(video_shape, kernel, dilation, heads, dim_head, causal=True, lo=None):
    """do the window kernels take this geometry (incl. the LDS their key-slot tables need) in the current precision mode?"""
    g = s3_geom(1, 2, video_shape, kernel, dilation, heads, dim_head, causal=causal)
    lo = (get_precision() != 'bf16') if lo is None else lo
    return bool(_lib.lib().amdnuwa_s3_supported(C.byref(g), 1 if lo else 0))


def s3_f16_supported(g):
    return bool(_lib.lib().amdnuwa_s3_f16_supported(C.byref(g)))


@_family('s3', _s3_work('fwd'))
def sparse3dna_fwd(g, qkv, wth, rel_bias=None, o_f16=False):
    """qkv: BF [B*ntok, 3*inner] (q | k | v);  returns o BF [B*ntok, inner].  rel_bias: fp32 [J, heads] or None.
    o_f16 (fp16 operand form only): o = BF(bf16 copy, None, fp16 copy) -- the operand of the two-MFMA to_out product"""
    L = _lib.lib()
    g.rel_bias, g.d_rel_bias = _p(rel_bias), None
    inner = g.heads * g.dim_head
    R = g.B * g.ntok
    if qkv.f16 is not None:            # fp16 operand form: single fp16 MFMAs; output hi + lo, or (o_f16) a bf16 copy + an fp16 copy
        q16, k16, v16 = (qkv.f16[:, i * inner:(i + 1) * inner] for i in range(3))
        dev = qkv.f16.device
        if o_f16 == 'only':            # ... or the fp16 copy alone (the block's backward runs on fp16 gradients: sparse3dna_bwd16)
            o = BF(None, None, torch.empty((R, inner), dtype=torch.float16, device=dev))
        elif o_f16:
            o = BF(torch.empty((R, inner), dtype=torch.bfloat16, device=dev), None,
                   torch.empty((R, inner), dtype=torch.float16, device=dev))
        else:
            o = empty_bf((R, inner), dev, lo=True)
        check(L.amdnuwa_sparse3dna_fwd_f16(C.byref(g), _p(q16), _p(k16), _p(v16), qkv.f16.stride(0), _p(wth), _p(o.hi),
                                           _p(o.f16 if o_f16 else o.lo), inner, 1 if o_f16 else 0, _stream()), 'amdnuwa_sparse3dna_fwd_f16')
        return o
    o = empty_bf((R, inner), qkv.hi.device, lo=qkv.lo is not None)
    q, k, v = (view(qkv, cols=slice(i * inner, (i + 1) * inner)) for i in range(3))
    check(L.amdnuwa_sparse3dna_fwd(C.byref(g), _p(q.hi), _p(k.hi), _p(v.hi), _p(q.lo), _p(k.lo), _p(v.lo), qkv.hi.stride(0),
                                   _p(wth), _p(o.hi), _p(o.lo), inner, _stream()), 'amdnuwa_sparse3dna_fwd')
    return o


def s3_bwd16_supported(g):
    return bool(_lib.lib().amdnuwa_sparse3dna_bwd_f16_supported(C.byref(g)))


@_family('s3', _s3_work('bwd'))
def sparse3dna_bwd16(g, qkv16, wth, dO16, s2):
    """fp16-gradient form: qkv16 fp16 [R, 3*inner] (the forward's operands), dO16 = fp16(S dO) [R, inner], s2 = device {S, 1 / S};
    returns (dqkv fp16 [R, 3*inner] = fp16(S dqkv), dw_th fp32 [h, h])"""
    L = _lib.lib()
    g.rel_bias, g.d_rel_bias = None, None
    inner = g.heads * g.dim_head
    R = g.B * g.ntok
    dev = qkv16.device
    dqkv = torch.empty((R, 3 * inner), dtype=torch.float16, device=dev)
    dwth = torch.empty((g.heads, g.heads), dtype=torch.float32, device=dev)
    q, k, v = (qkv16[:, i * inner:(i + 1) * inner] for i in range(3))
    dq, dk, dv = (dqkv[:, i * inner:(i + 1) * inner] for i in range(3))
    nb = L.amdnuwa_sparse3dna_bwd_workspace_bytes(C.byref(g))
    ws = workspace(nb, dev)
    check(L.amdnuwa_sparse3dna_bwd_f16(C.byref(g), _p(q), _p(k), _p(v), qkv16.stride(0), _p(wth), _p(dO16), dO16.stride(0),
                                       _p(dq), _p(dk), _p(dv), dqkv.stride(0), _p(dwth), 0, _p(s2), _p(ws), nb, _stream()),
          'amdnuwa_sparse3dna_bwd_f16')
    return dqkv, dwth


@_family('s3', _s3_work('bwd'))
def sparse3dna_bwd(g, qkv, wth, dO, rel_bias=None):
    """returns (dqkv BF [R, 3*inner], dw_th fp32 [h, h]) -- and d(rel_bias) fp32 [J, heads] as a third item when rel_bias is given
    (callers that pass rel_bias=None through the keyword get a 3-tuple with None)"""
    L = _lib.lib()
    drel = torch.empty_like(rel_bias) if rel_bias is not None else None
    g.rel_bias, g.d_rel_bias = _p(rel_bias), _p(drel)
    inner = g.heads * g.dim_head
    R = g.B * g.ntok
    dev = qkv.hi.device
    dqkv = empty_bf((R, 3 * inner), dev, lo=qkv.lo is not None)
    dwth = torch.empty((g.heads, g.heads), dtype=torch.float32, device=dev)
    q, k, v = (view(qkv, cols=slice(i * inner, (i + 1) * inner)) for i in range(3))
    dq, dk, dv = (view(dqkv, cols=slice(i * inner, (i + 1) * inner)) for i in range(3))
    nb = L.amdnuwa_sparse3dna_bwd_workspace_bytes(C.byref(g))
    ws = workspace(nb, dev)
    check(L.amdnuwa_sparse3dna_bwd(C.byref(g), _p(q.hi), _p(k.hi), _p(v.hi), _p(q.lo), _p(k.lo), _p(v.lo), qkv.hi.stride(0),
                                   _p(wth), _p(dO.hi), _p(dO.lo), dO.hi.stride(0), _p(dq.hi), _p(dk.hi), _p(dv.hi),
                                   _p(dq.lo), _p(dk.lo), _p(dv.lo), dqkv.hi.stride(0), _p(dwth), 0, _p(ws), nb, _stream()),
          'amdnuwa_sparse3dna_bwd')
    g.d_rel_bias = None
    return dqkv, dwth, drel


def cross2dna_fwd(g, q, kv, null_k, null_v, mask_u8, wth, ctx_rows):
    """SparseCross2DNA core.  q BF [B*ntok, inner]; kv BF [B*ctx_rows, 2*inner] (k | v); null_k / null_v BF [inner]; mask_u8 [B, ctx_rows]
    or None.  Returns o BF [B*ntok, inner] whose rows b*ntok (<bos>) are left for the caller."""
    L = _lib.lib()
    inner = g.heads * g.dim_head
    o = empty_bf((g.B * g.ntok, inner), q.hi.device, lo=q.lo is not None)
    k, v = (view(kv, cols=slice(i * inner, (i + 1) * inner)) for i in range(2))
    check(L.amdnuwa_cross2dna_fwd(C.byref(g), _p(q.hi), _p(q.lo), q.hi.stride(0), ctx_rows, _p(k.hi), _p(v.hi), _p(k.lo), _p(v.lo),
                                  kv.hi.stride(0), _p(null_k.hi), _p(null_k.lo), _p(null_v.hi), _p(null_v.lo), _p(mask_u8), _p(wth),
                                  _p(o.hi), _p(o.lo), inner, _stream()), 'amdnuwa_cross2dna_fwd')
    return o


def cross2dna_bwd(g, q, kv, null_k, null_v, mask_u8, wth, dO, ctx_rows):
    """returns dq BF [B*ntok, inner] (rows b*ntok untouched), dkv BF [B*ctx_rows, 2*inner], d_null_k, d_null_v fp32 [inner], dw_th [h, h]"""
    L = _lib.lib()
    inner = g.heads * g.dim_head
    dev = q.hi.device
    lo = q.lo is not None
    dq = empty_bf((g.B * g.ntok, inner), dev, lo=lo)
    dkv = empty_bf((g.B * ctx_rows, 2 * inner), dev, lo=lo)
    dnk = torch.empty(inner, dtype=torch.float32, device=dev)
    dnv = torch.empty(inner, dtype=torch.float32, device=dev)
    dwth = torch.empty((g.heads, g.heads), dtype=torch.float32, device=dev)
    k, v = (view(kv, cols=slice(i * inner, (i + 1) * inner)) for i in range(2))
    dk, dv = (view(dkv, cols=slice(i * inner, (i + 1) * inner)) for i in range(2))
    nb = L.amdnuwa_cross2dna_bwd_workspace_bytes(C.byref(g))
    ws = workspace(nb, dev)
    check(L.amdnuwa_cross2dna_bwd(C.byref(g), _p(q.hi), _p(q.lo), q.hi.stride(0), ctx_rows, _p(k.hi), _p(v.hi), _p(k.lo), _p(v.lo),
                                  kv.hi.stride(0), _p(null_k.hi), _p(null_k.lo), _p(null_v.hi), _p(null_v.lo), _p(mask_u8), _p(wth),
                                  _p(dO.hi), _p(dO.lo), dO.hi.stride(0), _p(dq.hi), _p(dq.lo), dq.hi.stride(0), _p(dk.hi), _p(dv.hi),
                                  _p(dk.lo), _p(dv.lo), dkv.hi.stride(0), _p(dnk), _p(dnv), _p(dwth), _p(ws), nb, _stream()),
          'amdnuwa_cross2dna_bwd')
    return dq, dkv, dnk, dnv, dwth


def decode_shift(h, cache, pos_dev, fmap):
    """h BF [B, D] (row pos of every sample) -> stored in cache BF [B, rows, D]; returns shift(h)[pos] as BF [B, D]"""
    L = _lib.lib()
    B, D = h.hi.shape
    out = empty_bf((B, D), h.hi.device, lo=h.lo is not None)
    check(L.amdnuwa_decode_shift(_p(h.hi), _p(h.lo), _p(cache.hi), _p(cache.lo), _p(out.hi), _p(out.lo), _p(pos_dev), B,
                                 cache.hi.shape[1], D, fmap, _stream()), 'amdnuwa_decode_shift')
    return out


def decode_ln(y, resid, post, nxt, cache=None, pos_dev=None, fmap=0, eps=1e-5):
    """the norms around a block for the single new row: post = (w, b) or None, nxt = (w, b) or None.
    returns (x_new fp32 [B, D] or None when resid is None, h BF [B, D] or None when nxt is None)"""
    L = _lib.lib()
    yp, ybf, (B, D), dev = _f32_or_bf(y)
    x_new = torch.empty((B, D), dtype=torch.float32, device=dev) if resid is not None else None
    h = empty_bf((B, D), dev) if nxt is not None else BF(None, None)
    ch, cl, rows = (cache.hi, cache.lo, cache.hi.shape[1]) if cache is not None else (None, None, 0)
    check(L.amdnuwa_decode_ln(yp, 1 if ybf else 0, _p(resid), _p(post[0]) if post else None, _p(post[1]) if post else None,
                              _p(nxt[0]) if nxt else None, _p(nxt[1]) if nxt else None, _p(x_new), _p(ch), _p(cl), _p(h.hi),
                              _p(h.lo), _p(pos_dev), B, rows, D, fmap, eps, _stream()), 'amdnuwa_decode_ln')
    return x_new, (h if nxt is not None else None)


def s3_decode(g, qkv, kv_cache, pos_dev, wth, rel_bias=None):
    """qkv BF [B, 3*inner] of the new row; kv_cache BF [B, rows, 2*inner]; returns o BF [B, inner]"""
    L = _lib.lib()
    g.rel_bias, g.d_rel_bias = _p(rel_bias), None
    inner = g.heads * g.dim_head
    o = empty_bf((g.B, inner), qkv.hi.device, lo=qkv.lo is not None)
    check(L.amdnuwa_s3_decode(C.byref(g), _p(qkv.hi), _p(qkv.lo), _p(kv_cache.hi), _p(kv_cache.lo), kv_cache.hi.shape[1],
                              _p(pos_dev), _p(wth), _p(o.hi), _p(o.lo), _stream()), 'amdnuwa_s3_decode')
    return o


def xattn_decode(g, q, pk, wth):
    """single-query text cross-attention (g.n == 1): q BF [B, inner] -> o BF [B, inner]"""
    L = _lib.lib()
    inner = g.heads * g.dim_head
    o = empty_bf((g.B, inner), q.hi.device, lo=q.lo is not None)
    check(L.amdnuwa_xattn_decode(C.byref(g), _p(q.hi), _p(q.lo), q.hi.stride(0), C.byref(pk.struct), _p(wth), _p(o.hi), _p(o.lo),
                                 inner, _stream()), 'amdnuwa_xattn_decode')
    return o


def x_geom(B, n, T, heads, dim_head):
    g = XGeom()
    g.B, g.n, g.T = B, n, T
    g.JP = _lib.lib().amdnuwa_xattn_jp(T)
    g.heads, g.dim_head, g.scale = heads, dim_head, dim_head ** -0.5
    return g


class PackedKV:
    """per-(sample, head) key/value images for the cross-attention kernels"""

    def __init__(self, g, device, lo, lean=False):
        """lean (fp16 lo images only): just the four images the fp16 forward core (K [key][d] and V^T in fp16) and the bf16 backward (K and V
        [key][d] in bf16) of the 'bf16x3-fwd' training step read; the other four are neither allocated nor written"""
        sh1 = (g.B, g.heads, g.JP, g.dim_head)
        sh2 = (g.B, g.heads, g.dim_head, g.JP)
        mk = lambda s: empty_bf(s, device, lo=lo)
        if lean == 'bwd':                        # the two bf16 [key][d] images of the recomputing backward alone (the forward runs on xattn6 images)
            e16 = lambda s: torch.empty(s, dtype=torch.bfloat16, device=device)
            self.Kp, self.Vp = BF(e16(sh1), None), BF(e16(sh1), None)
            self.Kt, self.Vt = BF(None, None), BF(None, None)
        elif lean:
            e16 = lambda s: torch.empty(s, dtype=torch.bfloat16, device=device)
            self.Kp, self.Vp = BF(e16(sh1), e16(sh1)), BF(e16(sh1), None)
            self.Kt, self.Vt = BF(None, None), BF(None, e16(sh2))
        else:
            self.Kp, self.Vp, self.Kt, self.Vt = mk(sh1), mk(sh1), mk(sh2), mk(sh2)
        self.valid = torch.empty((g.B, g.JP), dtype=torch.uint8, device=device)
        s = XKV()
        s.Kp, s.Kp_lo, s.Kt, s.Kt_lo = _p(self.Kp.hi), _p(self.Kp.lo), _p(self.Kt.hi), _p(self.Kt.lo)
        s.Vp, s.Vp_lo, s.Vt, s.Vt_lo = _p(self.Vp.hi), _p(self.Vp.lo), _p(self.Vt.hi), _p(self.Vt.lo)
        s.valid = _p(self.valid)
        self.struct = s
        self.f16 = False

    def drop_lo(self):
        """release the lo images (mixed mode: the bf16 backward reads the hi images only)"""
        self.Kp, self.Vp, self.Kt, self.Vt = (BF(t.hi, None) for t in (self.Kp, self.Vp, self.Kt, self.Vt))
        s = self.struct
        s.Kp_lo = s.Kt_lo = s.Vp_lo = s.Vt_lo = None
        return self


def xattn_pack(g, kv, null_k, null_v, mask_u8, out=None, lean=False):
    """kv with an f16 copy: the lo images of the result are FP16 images (for xattn2_fwd_f16), not bf16 residuals.
    out: a PackedKV of the same geometry / operand form to pack INTO (a captured HIP graph keeps reading the same buffers)"""
    L = _lib.lib()
    if out is not None:
        assert kv.f16 is None and (out.Kp.lo is not None) == (kv.lo is not None)
        check(L.amdnuwa_xattn_pack(C.byref(g), _p(kv.hi), _p(kv.lo), kv.hi.stride(0), _p(null_k), _p(null_v), _p(mask_u8),
                                   C.byref(out.struct), _stream()), 'amdnuwa_xattn_pack')
        return out
    if lean == 'bwd':
        pk = PackedKV(g, kv.hi.device, False, lean='bwd')
        check(L.amdnuwa_xattn_pack(C.byref(g), _p(kv.hi), None, kv.hi.stride(0), _p(null_k), _p(null_v), _p(mask_u8),
                                   C.byref(pk.struct), _stream()), 'amdnuwa_xattn_pack')
        return pk
    if kv.f16 is not None:
        pk = PackedKV(g, kv.hi.device, True, lean=lean)
        check(L.amdnuwa_xattn_pack_f16(C.byref(g), _p(kv.hi), _p(kv.f16), kv.hi.stride(0), _p(null_k), _p(null_v), _p(mask_u8),
                                       C.byref(pk.struct), _stream()), 'amdnuwa_xattn_pack_f16')
        pk.f16 = True
        return pk
    pk = PackedKV(g, kv.hi.device, kv.lo is not None)
    check(L.amdnuwa_xattn_pack(C.byref(g), _p(kv.hi), _p(kv.lo), kv.hi.stride(0), _p(null_k), _p(null_v), _p(mask_u8),
                               C.byref(pk.struct), _stream()), 'amdnuwa_xattn_pack')
    return pk


@_family('xattn', _x_work('fwd'))
def xattn2_fwd_f16(g, q, pk, wth, o_f16=False):
    """the xattn4 core on fp16 operands (q.f16, the fp16 images of pk): returns o BF [B*n, inner] (hi + lo; with o_f16 a bf16 copy + an
    fp16 copy, the operand of the two-MFMA to_out product) and the statistics"""
    L = _lib.lib()
    assert q.f16 is not None and getattr(pk, 'f16', False)
    inner = g.heads * g.dim_head
    dev = q.hi.device
    if o_f16:
        o = BF(torch.empty((g.B * g.n, inner), dtype=torch.bfloat16, device=dev), None,
               torch.empty((g.B * g.n, inner), dtype=torch.float16, device=dev))
    else:
        o = empty_bf((g.B * g.n, inner), dev, lo=True)
    stats = torch.empty((g.B, g.heads, g.n, 2), dtype=torch.float32, device=dev)
    check(L.amdnuwa_xattn2_fwd_f16(C.byref(g), _p(q.f16), q.f16.stride(0), C.byref(pk.struct), _p(wth), _p(o.hi),
                                   _p(o.f16 if o_f16 else o.lo), inner, 1 if o_f16 else 0, _p(stats), _stream()), 'amdnuwa_xattn2_fwd_f16')
    return o, stats


class PackedKV6:
    """key / value images of the xattn6 cross-attention kernels (amdnuwa_xattn6_pack): K6 / V6 [B, nch, heads, 32, 64] 16-bit in the
    kernels' LDS order, vbits [B, nch] (bit j of word c: context key 32 c + j takes part)"""

    def __init__(self, g, device, f16):
        L = _lib.lib()
        self.nch = L.amdnuwa_xattn6_nch(g.T)
        dt = torch.float16 if f16 else torch.bfloat16
        sh = (g.B, self.nch, g.heads, 32, g.dim_head)
        self.K6, self.V6 = torch.empty(sh, dtype=dt, device=device), torch.empty(sh, dtype=dt, device=device)
        self.vbits = torch.empty((g.B, self.nch), dtype=torch.int32, device=device)
        self.f16 = bool(f16)
        s = _lib.X6KV()
        s.K6, s.V6, s.vbits = _p(self.K6), _p(self.V6), _p(self.vbits)
        self.struct = s


_XATTN6 = os.environ.get('AMDNUWA_XATTN6', '1') != '0'


def set_xattn6(on):
    """the third-design cross-attention forward (amdnuwa_xattn6_*; AMDNUWA_XATTN6=0 keeps xattn4)"""
    global _XATTN6
    _XATTN6 = bool(on)


def xattn6_on():
    return _XATTN6


def xattn6_supported(g):
    return bool(_lib.lib().amdnuwa_xattn6_supported(C.byref(g)))


def xattn6_pack(g, kv16, mask_u8, out=None):
    """kv16: [B*T, ld] fp16 or bf16 tensor, keys in columns [0, inner), values in [inner, 2 inner) (= to_kv(context))"""
    f16 = kv16.dtype == torch.float16
    assert kv16.dtype in (torch.float16, torch.bfloat16) and kv16.stride(1) == 1
    pk = out if out is not None else PackedKV6(g, kv16.device, f16)
    assert pk.f16 == f16
    check(_lib.lib().amdnuwa_xattn6_pack(C.byref(g), _p(kv16), kv16.stride(0), _p(mask_u8), 1 if f16 else 0, C.byref(pk.struct), _stream()),
          'amdnuwa_xattn6_pack')
    return pk


@_family('xattn', _x_work('fwd'))
def xattn6_fwd(g, q16, pk, null_k, null_v, wth, o_f16=False, lo=True):
    """the xattn6 forward core on q16 [B*n, ld] (fp16 with fp16 images: every MFMA the fp16 one; else bf16).  Returns o BF [B*n, inner]
    (hi + lo pair; with o_f16 a bf16 copy + an fp16 copy; lo=False: the bf16 copy alone) and the softmax statistics [B, h, n, 2]"""
    L = _lib.lib()
    f16 = q16.dtype == torch.float16
    assert pk.f16 == f16 and q16.stride(1) == 1
    inner = g.heads * g.dim_head
    dev = q16.device
    if o_f16 == 'only':                # the fp16 copy alone (the block's backward runs on fp16 gradients: xattn6_bwd16)
        o = BF(None, None, torch.empty((g.B * g.n, inner), dtype=torch.float16, device=dev))
    elif o_f16:
        o = BF(torch.empty((g.B * g.n, inner), dtype=torch.bfloat16, device=dev), None,
               torch.empty((g.B * g.n, inner), dtype=torch.float16, device=dev))
    else:
        o = empty_bf((g.B * g.n, inner), dev, lo=lo)
    stats = torch.empty((g.B, g.heads, g.n, 2), dtype=torch.float32, device=dev)
    check(L.amdnuwa_xattn6_fwd(C.byref(g), _p(q16), q16.stride(0), C.byref(pk.struct), _p(null_k), _p(null_v), _p(wth), _p(o.hi),
                               _p(o.f16 if o_f16 else o.lo), inner, 1 if o_f16 else 0, _p(stats), 1 if f16 else 0, _stream()),
          'amdnuwa_xattn6_fwd')
    return o, stats


class PackedKV6B:
    """bf16 [key][d] images of the xattn6 backward (amdnuwa_xattn6_pack_bwd): key 0 = the null key, keys 1..T the context, JP / 32 chunks"""

    def __init__(self, g, device, f16=False):
        self.nch = g.JP // 32
        self.f16 = f16
        sh = (g.B, self.nch, g.heads, 32, g.dim_head)
        dt = torch.float16 if f16 else torch.bfloat16
        self.K6, self.V6 = torch.empty(sh, dtype=dt, device=device), torch.empty(sh, dtype=dt, device=device)
        self.vbits = torch.empty((g.B, self.nch), dtype=torch.int32, device=device)
        s = _lib.X6KV()
        s.K6, s.V6, s.vbits = _p(self.K6), _p(self.V6), _p(self.vbits)
        self.struct = s


def xattn6_bwd_ok(g):
    return _XATTN6 and os.environ.get('AMDNUWA_XATTN6_BWD', '1') != '0' and _lib.lib().amdnuwa_xattn6_bwd_image_bytes(C.byref(g)) > 0 and \
        xattn_chunk_major_ok(g)


def xattn6_pack_bwd(g, kv16, null_k, null_v, mask_u8):
    """kv16: the bf16 copy of to_kv(context) -- or its fp16 copy: the images of the fp16-gradient form (xattn6_bwd16)"""
    f16 = kv16.dtype == torch.float16
    assert (f16 or kv16.dtype == torch.bfloat16) and kv16.stride(1) == 1
    pk = PackedKV6B(g, kv16.device, f16=f16)
    pk.null_k, pk.null_v = null_k, null_v
    fn = _lib.lib().amdnuwa_xattn6_pack_bwd_f16 if f16 else _lib.lib().amdnuwa_xattn6_pack_bwd
    check(fn(C.byref(g), _p(kv16), kv16.stride(0), _p(null_k), _p(null_v), _p(mask_u8), C.byref(pk.struct), _stream()), 'amdnuwa_xattn6_pack_bwd')
    return pk


@_family('xattn', _x_work('bwd'))
def xattn6_bwd16(g, q16, dO16, pk, wth, stats, s2):
    """fp16-gradient form of xattn6_bwd: q16 fp16 [B*n, inner] (the forward's operand), dO16 = fp16(S dO), pk = fp16 images, s2 = device {S, 1 / S}.
    Returns dq fp16 (= S dq), dS fp16 (= S dS) and Pm fp16 chunk-major, dw_th fp32 [h, h] (unscaled)"""
    L = _lib.lib()
    assert pk.f16 and q16.dtype == torch.float16 and dO16.dtype == torch.float16
    inner = g.heads * g.dim_head
    dev = q16.device
    dq = torch.empty((g.B * g.n, inner), dtype=torch.float16, device=dev)
    shape = (g.B, g.heads, g.JP // 32, g.n, 32)
    dS, Pm = torch.empty(shape, dtype=torch.float16, device=dev), torch.empty(shape, dtype=torch.float16, device=dev)
    nb = L.amdnuwa_xattn6_bwd_workspace_bytes(C.byref(g))
    part = torch.empty((nb // (4 * g.heads * g.heads), g.heads * g.heads), dtype=torch.float32, device=dev)
    check(L.amdnuwa_xattn6_bwd_f16(C.byref(g), _p(q16), q16.stride(0), _p(dO16), dO16.stride(0), C.byref(pk.struct), _p(pk.null_k), _p(pk.null_v), _p(wth),
                                   _p(stats), _p(dS), _p(Pm), _p(dq), inner, _p(part), nb, _stream()), 'amdnuwa_xattn6_bwd_f16')
    dwth = (colsum(part) * (s2[1] * 64.0)).reshape(g.heads, g.heads)  # fixed-order reduction over the workgroups; the partials carry S / 64 (the V image's 2^-6)
    return dq, dS, Pm, dwth


@_family('xattn', _x_work('bwd'))
def xattn6_bwd(g, q, dO, pk, wth, stats):
    """the query side of the recomputing backward on xattn6 images: returns dq BF [B*n, inner], dS BF and Pm BF chunk-major
    [B, h, JP / 32, n, 32] (what xattn_kv_grads takes), dw_th fp32 [h, h] -- the results of xattn2_bwd(chunk_major=True)"""
    L = _lib.lib()
    inner = g.heads * g.dim_head
    dev = q.hi.device
    dq = empty_bf((g.B * g.n, inner), dev, lo=False)
    shape = (g.B, g.heads, g.JP // 32, g.n, 32)
    dS, Pm = empty_bf(shape, dev, lo=False), empty_bf(shape, dev, lo=False)
    nb = L.amdnuwa_xattn6_bwd_workspace_bytes(C.byref(g))
    part = torch.empty((nb // (4 * g.heads * g.heads), g.heads * g.heads), dtype=torch.float32, device=dev)
    check(L.amdnuwa_xattn6_bwd(C.byref(g), _p(q.hi), q.hi.stride(0), _p(dO.hi), dO.hi.stride(0), C.byref(pk.struct), _p(pk.null_k), _p(pk.null_v), _p(wth), _p(stats),
                               _p(dS.hi), _p(Pm.hi), _p(dq.hi), inner, _p(part), nb, _stream()), 'amdnuwa_xattn6_bwd')
    dwth = colsum(part).reshape(g.heads, g.heads)          # fixed-order reduction over the workgroups
    return dq, BF(dS.hi, None), BF(Pm.hi, None), dwth


def xattn_fwd(g, q, pk, wth, save=True, want_stats=False):
    """first-design kernel (bf16 or bf16x3).  save: keep P / P' for amdnuwa_xattn_bwd.  want_stats: return the softmax statistics
    [B, h, n, 2] the recomputing backward (xattn2_bwd) takes instead -- (o, stats) is then the result."""
    L = _lib.lib()
    inner = g.heads * g.dim_head
    dev = q.hi.device
    lo = q.lo is not None
    o = empty_bf((g.B * g.n, inner), dev, lo=lo)
    if save and not want_stats:
        P = empty_bf((g.B, g.heads, g.n, g.JP), dev, lo=lo)
        Pm = empty_bf((g.B, g.heads, g.n, g.JP), dev, lo=lo)
    else:
        P = Pm = BF(None, None)
    stats = torch.empty((g.B, g.heads, g.n, 2), dtype=torch.float32, device=dev) if want_stats else None
    check(L.amdnuwa_xattn_fwd_stats(C.byref(g), _p(q.hi), _p(q.lo), q.hi.stride(0), C.byref(pk.struct), _p(wth), _p(o.hi), _p(o.lo),
                                    inner, _p(P.hi), _p(P.lo), _p(Pm.hi), _p(Pm.lo), _p(stats), _stream()), 'amdnuwa_xattn_fwd_stats')
    if want_stats:
        return o, stats
    return o, P, Pm


def xattn_bwd(g, dO, pk, wth, P):
    """returns dq BF [B*n, inner], dS BF [B,h,n,JP], dw_th"""
    L = _lib.lib()
    inner = g.heads * g.dim_head
    dev = dO.hi.device
    lo = dO.lo is not None
    dq = empty_bf((g.B * g.n, inner), dev, lo=lo)
    dS = empty_bf((g.B, g.heads, g.n, g.JP), dev, lo=lo)
    dwth = torch.empty((g.heads, g.heads), dtype=torch.float32, device=dev)
    nb = L.amdnuwa_xattn_bwd_workspace_bytes(C.byref(g))
    ws = workspace(nb, dev)
    check(L.amdnuwa_xattn_bwd(C.byref(g), _p(dO.hi), _p(dO.lo), dO.hi.stride(0), C.byref(pk.struct), _p(wth), _p(P.hi), _p(P.lo),
                              _p(dS.hi), _p(dS.lo), _p(dq.hi), _p(dq.lo), inner, _p(dwth), 0, _p(ws), nb, _stream()),
          'amdnuwa_xattn_bwd')
    return dq, dS, dwth


def xattn2_supported(g, q=None):
    """second-design cross-attention kernels: fast bf16 mode (no lo parts), 8 heads x 64"""
    return (q is None or q.lo is None) and bool(_lib.lib().amdnuwa_xattn2_supported(C.byref(g)))


@_family('xattn', _x_work('fwd'))
def xattn2_fwd(g, q, pk, wth):
    """returns o BF [B*n, inner], stats fp32 [B, h, n, 2] = (row max, 1 / row sum) of the masked, scaled scores"""
    L = _lib.lib()
    inner = g.heads * g.dim_head
    dev = q.hi.device
    o = empty_bf((g.B * g.n, inner), dev, lo=False)
    stats = torch.empty((g.B, g.heads, g.n, 2), dtype=torch.float32, device=dev)
    check(L.amdnuwa_xattn2_fwd(C.byref(g), _p(q.hi), q.hi.stride(0), C.byref(pk.struct), _p(wth), _p(o.hi), inner, _p(stats),
                               _stream()), 'amdnuwa_xattn2_fwd')
    return o, stats


def _xattn_tn_desc(g, Mx, chunked):
    """the batched TN product of the cross attention's dK / dV (shape part only): A = dS or Pm per (sample, head), B = q or dO"""
    d = GemmDesc()
    d.lda, d.ldb = g.JP, g.heads * g.dim_head
    d.M, d.N, d.K = Mx, g.dim_head, g.n
    d.batch, d.batch_inner = g.B * g.heads, g.heads
    d.strideA_inner = g.n * g.JP                # A / C are dense over (b, h): inner stride = one head
    d.strideA = g.heads * g.n * g.JP
    d.strideC_inner = g.JP * g.dim_head
    d.strideC = g.heads * g.JP * g.dim_head
    d.ldc, d.c_is_bf16, d.beta = g.dim_head, 0, 0.0
    d.a_chunk32 = 1 if chunked else 0
    return d


def xattn_chunk_major_ok(g):
    """dS / Pm of xattn2_bwd chunk-major ([B, h, JP / 32, n, 32]: 1 KiB contiguous per store instruction of the kernel) -- when the batched
    TN product that reads them runs on the whole-M kernel, the one that takes that layout"""
    L = _lib.lib()
    if os.environ.get('AMDNUWA_XATTN_CM', '1') == '0':      # (env: A/B against the row-major arrays)
        return False
    return bool(L.amdnuwa_gemm_tn_chunked_a_supported(C.byref(_xattn_tn_desc(g, xattn_permuted_extent(g), True))))


def xattn_rows(g, t):
    """dS / Pm as [B, h, n, columns] whichever layout xattn2_bwd wrote (tests, tools)"""
    if t.dim() == 5:
        return t.permute(0, 1, 3, 2, 4).reshape(g.B, g.heads, g.n, g.JP)[..., :xattn_permuted_extent(g)]
    return t


@_family('xattn', _x_work('bwd'))
def xattn2_bwd(g, q, dO, pk, wth, stats, chunk_major=None):
    """returns dq BF [B*n, inner], dS BF and Pm BF, dw_th fp32 [h, h].  dS / Pm: [B, h, n, columns] (views of JP-pitch rows), or -- chunk_major,
    the default wherever xattn_chunk_major_ok -- [B, h, JP / 32, n, 32]; xattn_kv_grads takes either, xattn_rows shows either as rows"""
    L = _lib.lib()
    inner = g.heads * g.dim_head
    dev = q.hi.device
    cm = xattn_chunk_major_ok(g) if chunk_major is None else bool(chunk_major)
    dq = empty_bf((g.B * g.n, inner), dev, lo=False)
    shape = (g.B, g.heads, g.JP // 32, g.n, 32) if cm else (g.B, g.heads, g.n, g.JP)
    dS = empty_bf(shape, dev, lo=False)
    Pm = empty_bf(shape, dev, lo=False)
    nb = L.amdnuwa_xattn2_bwd_workspace_bytes(C.byref(g))
    part = torch.empty((nb // (4 * g.heads * g.heads), g.heads * g.heads), dtype=torch.float32, device=dev)
    check(L.amdnuwa_xattn2_bwd_ex(C.byref(g), _p(q.hi), q.hi.stride(0), _p(dO.hi), dO.hi.stride(0), C.byref(pk.struct), _p(wth),
                                  _p(stats), _p(dS.hi), _p(Pm.hi), _p(dq.hi), inner, _p(part), nb, 1 if cm else 0, _stream()),
          'amdnuwa_xattn2_bwd_ex')
    dwth = colsum(part).reshape(g.heads, g.heads)          # fixed-order reduction over the workgroups
    if cm:
        return dq, BF(dS.hi, None), BF(Pm.hi, None), dwth
    # lane groups of the last chunk whose 8 keys are all padding write nothing: hand out the columns that exist (views: the row pitch stays JP)
    mx = xattn_permuted_extent(g)
    return dq, BF(dS.hi[..., :mx], None), BF(Pm.hi[..., :mx], None), dwth


def xattn_permuted_extent(g):
    """columns of the chunk-permuted dS / Pm arrays (rows of dKp / dVp) that hold a key: the kernel's lane group g4 of chunk ch owns keys
    32 ch + {4 g4 .. 4 g4 + 3, 16 + 4 g4 .. 16 + 4 g4 + 3} at positions 32 ch + 8 g4 .. + 7, and key T is the last one"""
    ch, kk = g.T // 32, g.T % 32
    return 32 * ch + 8 * (min(kk // 4, 3) + 1)


_XATTN_RC = os.environ.get('AMDNUWA_XATTN_RC', '0') == '1'


def set_xattn_rc(on):
    """the recomputing key side of the cross-attention backward (amdnuwa_xattn2_bwd_rc: no dS / Pm arrays, 3 GB less workspace and
    HBM traffic per layer call at b = 128).  Off by default: the kernel is bound by LDS operand reads and the step measured 1.2 %
    slower with it (558 vs 551 ms); env AMDNUWA_XATTN_RC=1 turns it on."""
    global _XATTN_RC
    _XATTN_RC = bool(on)


def xattn2_bwd_rc_ok(g):
    return _XATTN_RC and bool(_lib.lib().amdnuwa_xattn2_bwd_rc_supported(C.byref(g)))


def xattn2_bwd_rc(g, q, dO, pk, wth, stats):
    """query side + recomputing key side: returns dq BF [B*n, inner], dKp / dVp fp32 [B, h, JP, dh], dw_th fp32 [h, h]"""
    L = _lib.lib()
    inner = g.heads * g.dim_head
    dev = q.hi.device
    dq = empty_bf((g.B * g.n, inner), dev, lo=False)
    nb = L.amdnuwa_xattn2_bwd_workspace_bytes(C.byref(g))
    part = torch.empty((nb // (4 * g.heads * g.heads), g.heads * g.heads), dtype=torch.float32, device=dev)
    sb = L.amdnuwa_xattn2_bwd_rc_stats_bytes(C.byref(g))
    nbd = torch.empty(sb // 4, dtype=torch.float32, device=dev)
    dKp = torch.empty((g.B, g.heads, g.JP, g.dim_head), dtype=torch.float32, device=dev)
    dVp = torch.empty_like(dKp)
    check(L.amdnuwa_xattn2_bwd_rc(C.byref(g), _p(q.hi), q.hi.stride(0), _p(dO.hi), dO.hi.stride(0), C.byref(pk.struct), _p(wth),
                                  _p(stats), _p(dq.hi), inner, _p(part), nb, _p(nbd), sb, _p(dKp), _p(dVp), _stream()),
          'amdnuwa_xattn2_bwd_rc')
    return dq, dKp, dVp, colsum(part).reshape(g.heads, g.heads)


@_family('xattn', _x_work('kv'))
def xattn_kv_grads16(g, dS16, Pm16, q16, dO16, s2):
    """fp16-gradient form of xattn_kv_grads on the chunk-major fp16 arrays of xattn6_bwd16: dKp = scale / S * (S dS)^T q, dVp = 1 / S * Pm^T (S dO)"""
    dev = q16.device
    Mx = xattn_permuted_extent(g)
    dKp = torch.empty((g.B, g.heads, g.JP, g.dim_head), dtype=torch.float32, device=dev)
    dVp = torch.empty_like(dKp)
    for (A, Bm, out, alpha) in ((dS16, q16, dKp, g.scale), (Pm16, dO16, dVp, 1.0)):
        d = _xattn_tn_desc(g, Mx, True)
        d.A, d.B, d.ldb, d.ab_f16 = _p(A), _p(Bm), Bm.stride(0), 1
        d.strideB, d.strideB_inner = g.n * Bm.stride(0), g.dim_head
        d.C, d.alpha, d.alpha_dev = _p(out), float(alpha), s2.data_ptr() + 4
        gemm_tn_batched(d, dev)
    return dKp, dVp


def xattn_bwd16_ok(g):
    """do the fp16-gradient kernels of the cross attention take this geometry (xattn6 backward + the whole-M TN kernel on fp16 chunk-major arrays)?"""
    if not xattn6_bwd_ok(g):
        return False
    d = _xattn_tn_desc(g, xattn_permuted_extent(g), True)
    d.A, d.B, d.C, d.ab_f16 = ctypes_dummy(), ctypes_dummy(), ctypes_dummy(), 1
    d.ldb = g.heads * g.dim_head
    L = _lib.lib()
    return bool(L.amdnuwa_gemm_tn_f16_supported(C.byref(d))) and bool(L.amdnuwa_gemm_tn_chunked_a_supported(C.byref(d)))


@_family('xattn', _x_work('kv'))
def xattn_kv_grads(g, dS, Pm, q, dO):
    """dKp = scale * dS^T q, dVp = Pm^T dO per (sample, head): two batched TN GEMMs (reduction over queries).
    returns fp32 [B, h, JP, dh] x 2 (rows past the last column of dS / Pm -- padding keys -- are not written)"""
    dev = q.hi.device
    cm = dS.hi.dim() == 5                      # chunk-major dS / Pm (xattn2_bwd)
    Mx = xattn_permuted_extent(g) if cm else dS.hi.shape[-1]      # (xattn2_bwd hands out only the columns that hold a key)
    dKp = torch.empty((g.B, g.heads, g.JP, g.dim_head), dtype=torch.float32, device=dev)
    dVp = torch.empty_like(dKp)
    for (A, Bm, out, alpha) in ((dS, q, dKp, g.scale), (Pm, dO, dVp, 1.0)):
        x3 = A.lo is not None and Bm.lo is not None
        d = _xattn_tn_desc(g, Mx, cm)
        d.A, d.Alo = _p(A.hi), _p(A.lo) if x3 else None
        d.B, d.Blo, d.ldb = _p(Bm.hi), _p(Bm.lo) if x3 else None, Bm.hi.stride(0)
        d.strideB, d.strideB_inner = g.n * Bm.hi.stride(0), g.dim_head
        d.C, d.alpha = _p(out), float(alpha)
        gemm_tn_batched(d, dev)
    return dKp, dVp


def xattn_key_positions(JP, device):
    """index tensor pos with  natural_order = permuted.index_select(dim, pos):  where key j of the plain order sits in the chunk-permuted
    order of xattn2_bwd's dS / Pm columns (and of the dKp / dVp rows computed from them)"""
    j = torch.arange(JP, device=device)
    kk = j & 31
    return (j & ~31) + 8 * ((kk & 15) >> 2) + 4 * (kk >> 4) + (kk & 3)


def xattn_unpack(g, dKp, dVp, lo, permuted=False, null_last=False):
    """permuted: dKp / dVp came from xattn_kv_grads over the dS / Pm of xattn2_bwd (chunk-permuted key rows); null_last: ... of xattn6_bwd
    (context key t at position t, the null key at position T)"""
    L = _lib.lib()
    inner = g.heads * g.dim_head
    dev = dKp.device
    dkv = empty_bf((g.B * g.T, 2 * inner), dev, lo=lo)
    dnk = torch.empty((g.heads, g.dim_head), dtype=torch.float32, device=dev)
    dnv = torch.empty_like(dnk)
    check(L.amdnuwa_xattn_unpack(C.byref(g), _p(dKp), _p(dVp), _p(dkv.hi), _p(dkv.lo), 2 * inner, _p(dnk), _p(dnv), (2 if permuted else 0) | (4 if null_last else 0), _stream()),
          'amdnuwa_xattn_unpack')
    return dkv, dnk, dnv


# ---- frozen VQGanVAE tokenizer (exact fp32) ----------------------------------------------------------------------

def _f32c(t):
    assert t.is_cuda, 'libamdnuwa kernels take device tensors'
    return t.detach().to(torch.float32).contiguous()


def conv2d_fwd(x, w, bias=None, stride=1, padding=0, leaky=False):
    """nn.Conv2d forward [+ LeakyReLU(0.1)], NCHW fp32 (reference vqgan_vae.py:352-365)."""
    L = _lib.lib()
    x, w = _f32c(x), _f32c(w)
    bias = _f32c(bias) if bias is not None else None
    N, Cin, H, W = x.shape
    Cout, Cin2, KH, KW = w.shape
    assert Cin == Cin2, (x.shape, w.shape)
    d = _lib.ConvDesc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.KH, d.KW, d.stride, d.pad = N, Cin, H, W, Cout, KH, KW, stride, padding
    d.Ho, d.Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    d.leaky = int(leaky)
    y = torch.empty((N, Cout, d.Ho, d.Wo), dtype=torch.float32, device=x.device)
    check(L.amdnuwa_conv2d_fwd(C.byref(d), _p(x), _p(w), _p(bias), _p(y), _stream()), 'amdnuwa_conv2d_fwd')
    return y


def groupnorm_fwd(x, w, b, groups, eps=1e-5, leaky=False):
    L = _lib.lib()
    x, w, b = _f32c(x), _f32c(w), _f32c(b)
    N, Cc = x.shape[:2]
    y = torch.empty_like(x)
    check(L.amdnuwa_groupnorm_fwd(_p(x), _p(w), _p(b), _p(y), N, Cc, x[0, 0].numel(), groups, eps, int(leaky), _stream()),
          'amdnuwa_groupnorm_fwd')
    return y


def vq_argmax(x, codebook, want_sim=False):
    """x [R, Dc], codebook [Cn, Dc] fp32 -> int64 indices [R] (cosine similarity, lowest index on ties)."""
    L = _lib.lib()
    x, codebook = _f32c(x), _f32c(codebook)
    R, Dc = x.shape
    idx = torch.empty((R,), dtype=torch.int64, device=x.device)
    sim = torch.empty((R,), dtype=torch.float32, device=x.device) if want_sim else None
    nb = L.amdnuwa_vq_argmax_workspace_bytes(R, codebook.shape[0])
    ws = workspace(nb, x.device)
    check(L.amdnuwa_vq_argmax_ws(_p(x), _p(codebook), _p(idx), _p(sim), R, codebook.shape[0], Dc, _p(ws), nb, _stream()), 'amdnuwa_vq_argmax_ws')
    return (idx, sim) if want_sim else idx


def vqgan_attention(x, qkv_w, out_w, out_b, bias, scale, ln_g, ln_b, heads, eps):
    """VQGanAttention.forward (reference vqgan_vae.py:263-286) on NCHW fp32 x: returns post_norm(to_out(attn)) + x.
    bias: continuous-position bias [heads, P, P] (parameters only); scale: the learned log-scale [heads]."""
    L = _lib.lib()
    x = _f32c(x)
    N, Cc, H, W = x.shape
    P_ = H * W
    qkv = conv2d_fwd(x, qkv_w, None, 1, 0)                       # [N, 3*heads*c, H, W]
    c = qkv.shape[1] // (3 * heads)
    # q and k = the first 2*heads*c of the 3*heads*c channel rows of every image
    check(L.amdnuwa_rows_l2norm(_p(qkv), N, 2 * heads * c, 3 * heads * c, P_, _stream()), 'amdnuwa_rows_l2norm')
    out = torch.empty((N, heads * c, H, W), dtype=torch.float32, device=x.device)
    check(L.amdnuwa_vqattn_core(_p(qkv), _p(_f32c(bias)), _p(_f32c(scale).reshape(-1)), _p(out), N, heads, c, P_, _stream()),
          'amdnuwa_vqattn_core')
    o = conv2d_fwd(out, out_w, out_b, 1, 0)
    y = torch.empty_like(o)
    check(L.amdnuwa_chan_layernorm(_p(o), _p(_f32c(ln_g).reshape(-1)), _p(_f32c(ln_b).reshape(-1)), _p(x), _p(y), N, o.shape[1], P_,
                                   float(eps), _stream()), 'amdnuwa_chan_layernorm')
    return y


def glu_chan(x):
    """nn.GLU(dim=1) on NCHW fp32"""
    L = _lib.lib()
    x = _f32c(x)
    N, C2, H, W = x.shape
    y = torch.empty((N, C2 // 2, H, W), dtype=torch.float32, device=x.device)
    check(L.amdnuwa_glu_chan(_p(x), _p(y), N, C2 // 2, H * W, _stream()), 'amdnuwa_glu_chan')
    return y


def upsample_bilinear2x(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) on NCHW fp32"""
    L = _lib.lib()
    x = _f32c(x)
    N, Cc, H, W = x.shape
    y = torch.empty((N, Cc, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    check(L.amdnuwa_upsample_bilinear2x(_p(x), _p(y), N, Cc, H, W, _stream()), 'amdnuwa_upsample_bilinear2x')
    return y
