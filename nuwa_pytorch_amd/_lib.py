"""ctypes binding of libamdnuwa.so (include/amdnuwa.h).  There is NO fallback: if the library is
missing or a call fails, a RuntimeError is raised -- the product path never silently degrades to
PyTorch eager or to the oracle."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# AMDNUWA_LIBRARY: another build of the same library (A/B runs of compiler options inside one process group; tools/ only)
LIB_PATH = os.environ.get('AMDNUWA_LIBRARY') or os.path.join(HERE, 'lib', 'libamdnuwa.so')
ABI_VERSION = 19

P = C.c_void_p
I = C.c_int
LL = C.c_longlong
F = C.c_float
SZ = C.c_size_t


class GemmDesc(C.Structure):
    _fields_ = [('A', P), ('Alo', P), ('strideA', LL), ('lda', I),
                ('B', P), ('Blo', P), ('strideB', LL), ('ldb', I),
                ('C', P), ('Clo', P), ('strideC', LL), ('ldc', I),
                ('c_is_bf16', I), ('bias', P), ('alpha', F), ('beta', F),
                ('M', I), ('N', I), ('K', I), ('batch', I), ('shift_ntok', I), ('shift_fmap', I),
                ('batch_inner', I), ('strideA_inner', LL), ('strideB_inner', LL), ('strideC_inner', LL),
                ('C2', P), ('C2lo', P), ('ldc2', I), ('geglu_u', P), ('geglu_u_lo', P), ('ld_u', I), ('c_lo_f16', I), ('ab_f16', I), ('c_f16', I), ('alpha_dev', P), ('a_chunk32', I)]


class S3Geom(C.Structure):
    _fields_ = [('B', I), ('ntok', I), ('F', I), ('H', I), ('W', I), ('kf', I), ('kh', I), ('kw', I),
                ('df', I), ('dh', I), ('dw', I), ('heads', I), ('dim_head', I), ('scale', F),
                ('rel_bias', P), ('d_rel_bias', P), ('noncausal', I)]


class XGeom(C.Structure):
    _fields_ = [('B', I), ('n', I), ('T', I), ('JP', I), ('heads', I), ('dim_head', I), ('scale', F)]


class XKV(C.Structure):
    _fields_ = [('Kp', P), ('Kp_lo', P), ('Kt', P), ('Kt_lo', P), ('Vp', P), ('Vp_lo', P), ('Vt', P), ('Vt_lo', P),
                ('valid', P)]


class X6KV(C.Structure):
    _fields_ = [('K6', P), ('V6', P), ('vbits', P)]


class ConvDesc(C.Structure):
    _fields_ = [('N', I), ('Cin', I), ('H', I), ('W', I), ('Cout', I), ('KH', I), ('KW', I), ('stride', I), ('pad', I),
                ('Ho', I), ('Wo', I), ('leaky', I)]


GD, SG, XG, XK, CD, X6 = (C.POINTER(t) for t in (GemmDesc, S3Geom, XGeom, XKV, ConvDesc, X6KV))

# name -> (restype, argtypes).  Mirrors include/amdnuwa.h declaration by declaration.
SIGNATURES = {
    'amdnuwa_abi_version': (I, []),
    'amdnuwa_error_string': (C.c_char_p, [I]),
    'amdnuwa_set_tuning': (I, [I, I]),
    'amdnuwa_get_tuning': (I, [I]),
    'amdnuwa_f16_sat_count': (C.c_ulonglong, [I]),
    'amdnuwa_timer_arm': (None, [I]),
    'amdnuwa_timer_begin': (I, [P]),
    'amdnuwa_timer_end': (I, [P]),
    'amdnuwa_timer_collect': (I, [C.POINTER(C.c_double), C.POINTER(LL)]),
    'amdnuwa_timer_collect_each': (I, [C.POINTER(C.c_double), LL, C.POINTER(LL)]),
    'amdnuwa_gemm_nt': (I, [GD, P]),
    'amdnuwa_gemm_tn_f16_supported': (I, [GD]),
    'amdnuwa_gemm_tn_chunked_a_supported': (I, [GD]),
    'amdnuwa_gemm_tn_workspace_bytes': (SZ, [GD]),
    'amdnuwa_gemm_tn': (I, [GD, P, SZ, P]),
    'amdnuwa_ln_fwd': (I, [P, P, P, P, P, P, P, P, P, P, LL, I, I, I, F, I, I, P]),
    'amdnuwa_ln_bwd_chain_workspace_bytes': (SZ, [LL, I]),
    'amdnuwa_ln_bwd_chain': (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, LL, I, I, I, I, P, SZ, P]),
    'amdnuwa_ln_bwd_chain_f16': (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, LL, I, I, I, I, I, P, P, SZ, P]),
    'amdnuwa_ln_post_pre_fwd': (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, LL, I, I, F, I, I, P]),
    'amdnuwa_ln_bwd_workspace_bytes': (SZ, [LL, I]),
    'amdnuwa_ln_bwd': (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, LL, I, I, I, I, I, P, SZ, P]),
    'amdnuwa_ln_bwd_f16': (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, LL, I, I, I, I, I, P, P, SZ, P]),
    'amdnuwa_colsum_workspace_bytes': (SZ, [LL, I]),
    'amdnuwa_colsum': (I, [P, P, LL, I, I, P, SZ, P]),
    'amdnuwa_geglu_fwd': (I, [P, P, P, P, LL, I, P]),
    'amdnuwa_gemm_nt_fused': (I, [GD]),
    'amdnuwa_gemm_nt_f16_fused': (I, [GD]),
    'amdnuwa_gemm_nt_f16ops_supported': (I, [GD]),
    'amdnuwa_gemm_nt_f16x2_supported': (I, [GD]),
    'amdnuwa_hilo_to_f16': (I, [P, P, I, P, I, LL, I, P]),
    'amdnuwa_geglu_il_fwd': (I, [P, P, P, P, LL, I, P]),
    'amdnuwa_geglu_il_bwd': (I, [P, P, P, P, P, P, LL, I, P]),
    'amdnuwa_geglu_bwd': (I, [P, P, P, P, P, P, LL, I, P]),
    'amdnuwa_cast_pad': (I, [P, I, P, P, I, LL, I, I, P]),
    'amdnuwa_transpose_cast': (I, [P, I, P, P, I, I, I, P]),
    'amdnuwa_embed_fwd': (I, [P, P, P, P, P, P, P, I, I, I, I, I, F, P]),
    'amdnuwa_embed_bwd_workspace_bytes': (SZ, [I, I]),
    'amdnuwa_embed_bwd': (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P, SZ, P]),
    'amdnuwa_ce_fwd': (I, [P, P, P, P, P, P, LL, I, I, F, P]),
    'amdnuwa_scale_by_device_scalar': (I, [P, SZ, P, P]),
    'amdnuwa_linear_ce_workspace_bytes': (SZ, [LL, I]),
    'amdnuwa_linear_ce': (I, [P, I, P, I, P, LL, I, I, F, P, P, P, I, P, SZ, P]),
    'amdnuwa_linear_ce_x3': (I, [P, P, P, I, P, P, P, I, P, LL, I, I, F, P, P, P, I, P, SZ, P]),
    'amdnuwa_s3_supported': (I, [SG, I]),
    'amdnuwa_sparse3dna_fwd': (I, [SG, P, P, P, P, P, P, I, P, P, P, I, P]),
    'amdnuwa_s3_f16_supported': (I, [SG]),
    'amdnuwa_sparse3dna_fwd_f16': (I, [SG, P, P, P, I, P, P, P, I, I, P]),
    'amdnuwa_sparse3dna_bwd_workspace_bytes': (SZ, [SG]),
    'amdnuwa_sparse3dna_bwd': (I, [SG, P, P, P, P, P, P, I, P, P, P, I, P, P, P, P, P, P, I, P, I, P, SZ, P]),
    'amdnuwa_sparse3dna_bwd_f16_supported': (I, [SG]),
    'amdnuwa_sparse3dna_bwd_f16': (I, [SG, P, P, P, I, P, P, I, P, P, P, I, P, I, P, P, SZ, P]),
    'amdnuwa_cross2dna_fwd': (I, [SG, P, P, I, I, P, P, P, P, I, P, P, P, P, P, P, P, P, I, P]),
    'amdnuwa_cross2dna_bwd_workspace_bytes': (SZ, [SG]),
    'amdnuwa_cross2dna_bwd': (I, [SG, P, P, I, I, P, P, P, P, I, P, P, P, P, P, P, P, P, I, P, P, I, P, P, P, P, I, P, P, P, P, SZ, P]),
    'amdnuwa_decode_shift': (I, [P, P, P, P, P, P, P, I, I, I, I, P]),
    'amdnuwa_decode_ln': (I, [P, I, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, P]),
    'amdnuwa_s3_decode': (I, [SG, P, P, P, P, I, P, P, P, P, P]),
    'amdnuwa_xattn_decode': (I, [XG, P, P, I, XK, P, P, P, I, P]),
    'amdnuwa_xattn_jp': (I, [I]),
    'amdnuwa_xattn_pack': (I, [XG, P, P, I, P, P, P, XK, P]),
    'amdnuwa_xattn_pack_f16': (I, [XG, P, P, I, P, P, P, XK, P]),
    'amdnuwa_xattn_fwd': (I, [XG, P, P, I, XK, P, P, P, I, P, P, P, P, P]),
    'amdnuwa_xattn_fwd_stats': (I, [XG, P, P, I, XK, P, P, P, I, P, P, P, P, P, P]),
    'amdnuwa_xattn_bwd_workspace_bytes': (SZ, [XG]),
    'amdnuwa_xattn_bwd': (I, [XG, P, P, I, XK, P, P, P, P, P, P, P, I, P, I, P, SZ, P]),
    'amdnuwa_xattn_unpack': (I, [XG, P, P, P, P, I, P, P, I, P]),
    'amdnuwa_xattn2_supported': (I, [XG]),
    'amdnuwa_xattn2_fwd': (I, [XG, P, I, XK, P, P, I, P, P]),
    'amdnuwa_xattn2_fwd_f16': (I, [XG, P, I, XK, P, P, P, I, I, P, P]),
    'amdnuwa_xattn2_bwd_workspace_bytes': (SZ, [XG]),
    'amdnuwa_xattn2_bwd': (I, [XG, P, I, P, I, XK, P, P, P, P, P, I, P, SZ, P]),
    'amdnuwa_xattn2_bwd_ex': (I, [XG, P, I, P, I, XK, P, P, P, P, P, I, P, SZ, I, P]),
    'amdnuwa_xattn6_supported': (I, [XG]),
    'amdnuwa_xattn6_nch': (I, [I]),
    'amdnuwa_xattn6_image_bytes': (SZ, [XG]),
    'amdnuwa_xattn6_pack': (I, [XG, P, I, P, I, X6, P]),
    'amdnuwa_xattn6_fwd': (I, [XG, P, I, X6, P, P, P, P, P, I, I, P, I, P]),
    'amdnuwa_xattn6_bwd_image_bytes': (SZ, [XG]),
    'amdnuwa_xattn6_pack_bwd': (I, [XG, P, I, P, P, P, X6, P]),
    'amdnuwa_xattn6_bwd_workspace_bytes': (SZ, [XG]),
    'amdnuwa_xattn6_bwd': (I, [XG, P, I, P, I, X6, P, P, P, P, P, P, P, I, P, SZ, P]),
    'amdnuwa_xattn6_pack_bwd_f16': (I, [XG, P, I, P, P, P, X6, P]),
    'amdnuwa_xattn6_bwd_f16': (I, [XG, P, I, P, I, X6, P, P, P, P, P, P, P, I, P, SZ, P]),
    'amdnuwa_xattn2_bwd_rc_supported': (I, [XG]),
    'amdnuwa_xattn2_bwd_rc_stats_bytes': (SZ, [XG]),
    'amdnuwa_xattn2_bwd_rc': (I, [XG, P, I, P, I, XK, P, P, P, I, P, SZ, P, SZ, P, P, P]),
    'amdnuwa_conv2d_fwd': (I, [CD, P, P, P, P, P]),
    'amdnuwa_groupnorm_fwd': (I, [P, P, P, P, I, I, I, I, F, I, P]),
    'amdnuwa_vq_argmax': (I, [P, P, P, P, LL, I, I, P]),
    'amdnuwa_vq_argmax_workspace_bytes': (SZ, [LL, I]),
    'amdnuwa_vq_argmax_ws': (I, [P, P, P, P, LL, I, I, P, SZ, P]),
    'amdnuwa_glu_chan': (I, [P, P, I, I, I, P]),
    'amdnuwa_upsample_bilinear2x': (I, [P, P, I, I, I, I, P]),
    'amdnuwa_grad_norm': (I, [P, I, F, P, P, P]),
    'amdnuwa_scale_grads': (I, [P, I, P, P]),
    'amdnuwa_adamw_step': (I, [P, I, F, F, F, F, P, P]),
    'amdnuwa_rows_l2norm': (I, [P, I, I, I, I, P]),
    'amdnuwa_vqattn_core': (I, [P, P, P, P, I, I, I, I, P]),
    'amdnuwa_chan_layernorm': (I, [P, P, P, P, P, I, I, I, F, P]),
    'amdnuwa_comm_available': (I, []),
    'amdnuwa_comm_last_error': (C.c_char_p, []),
    'amdnuwa_comm_unique_id': (I, [P, SZ]),
    'amdnuwa_comm_init': (I, [P, P, SZ, I, I, I]),
    'amdnuwa_comm_rank': (I, [P]),
    'amdnuwa_comm_world': (I, [P]),
    'amdnuwa_comm_allreduce': (I, [P, P, SZ, I, P]),
    'amdnuwa_comm_reduce_scatter_allgather': (I, [P, P, SZ, I, P]),
    'amdnuwa_comm_broadcast': (I, [P, P, SZ, I, P]),
    'amdnuwa_comm_destroy': (I, [P]),
}

_lib = None


def lib():
    """the loaded library; raises RuntimeError (loudly) when it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'libamdnuwa.so not found at {LIB_PATH}. Build it with `python -m nuwa_pytorch_amd.build` '
            '(hipcc --offload-arch=gfx950). nuwa_pytorch_amd has no PyTorch/CPU fallback for the hot path.')
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError(f'failed to load {LIB_PATH}: {e}') from e
    missing = [n for n in SIGNATURES if not hasattr(l, n)]
    if missing:
        raise RuntimeError(f'{LIB_PATH} lacks symbols declared in include/amdnuwa.h: {missing}')
    for n, (res, args) in SIGNATURES.items():
        fn = getattr(l, n)
        fn.restype = res
        fn.argtypes = args
    v = l.amdnuwa_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f'libamdnuwa ABI version {v} != expected {ABI_VERSION}: rebuild the library')
    # optional runtime tuning overrides, e.g. AMDNUWA_TUNING="0=2,6=1" (see amdnuwa_set_tuning in the header)
    for kv in filter(None, os.environ.get('AMDNUWA_TUNING', '').split(',')):
        k, v = kv.split('=')
        l.amdnuwa_set_tuning(int(k), int(v))
    _lib = l
    return l


def check(rc, what):
    if rc != 0:
        msg = lib().amdnuwa_error_string(int(rc))
        raise RuntimeError(f'{what} failed with code {rc}: {msg.decode() if msg else "?"}')
