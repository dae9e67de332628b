#!/usr/bin/env python
"""Bit-reproducibility of every kernel family of the training step at batches that fill the chip (several workgroups per CU, several
rounds): every kernel is run REP times on the same inputs and every output compared with the first run's, element for element.  The
full-size determinism tests of the GPU suite use one sample (one workgroup per CU at most); a race that needs co-resident workgroups only
shows up here.

    python tools/determinism_stress.py [B = 16] [--rep 8] [--only-s3]        (round-5 evidence: B = 128, --rep 10)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402

L = _lib.lib()
DEV = 'cuda'
heads, dh = 8, 64
inner = heads * dh
REP = int(sys.argv[sys.argv.index('--rep') + 1]) if '--rep' in sys.argv else 8


def tensors(o):
    out = []
    for x in (o if isinstance(o, (tuple, list)) else (o,)):
        if isinstance(x, K.BF):
            out += [t for t in (x.hi, x.lo, x.f16) if t is not None]
        elif isinstance(x, K.G16):
            out.append(x.t)
        elif torch.is_tensor(x):
            out.append(x)
    return out


def check(name, fn):
    ref = [t.clone() for t in tensors(fn())]
    bad = 0
    for _ in range(REP - 1):
        for a, b in zip(tensors(fn()), ref):
            bad += int((a != b).sum()) if a.dtype != torch.float32 else int((a.view(torch.int32) != b.view(torch.int32)).sum())
    print(f'{name:70s} {"bit-identical" if bad == 0 else f"{bad} ELEMENTS DIFFER"} over {REP} runs', flush=True)
    return bad


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
    only_s3 = '--only-s3' in sys.argv           # (the build-variant experiment of round 5: the 3DNA kernels alone)
    print('library:', _lib.LIB_PATH, flush=True)
    torch.manual_seed(0)
    total = 0
    wth = (torch.randn(heads, heads) * 0.5 + torch.eye(heads)).to(DEV)
    shape, kern = (10, 16, 16), (5, 3, 3)
    n = 2560
    for dil in ((1, 1, 1), (2, 2, 2), (4, 4, 4)):
        qkv = torch.randn(B * n, 3 * inner, device=DEV)
        g = K.s3_geom(B, n, shape, kern, dil, heads, dh)
        pbf = K.BF(qkv.to(torch.bfloat16), None)
        p16 = K.BF(qkv.to(torch.bfloat16), None, qkv.half())
        dO = K.BF(torch.randn(B * n, inner, device=DEV).to(torch.bfloat16), None)
        for rows in (1, 2):
            L.amdnuwa_set_tuning(16, rows)
            total += check(f'3DNA fwd bf16, dilation {dil[0]}, {rows} row(s) per tile, b={B}', lambda: K.sparse3dna_fwd(g, pbf, wth))
            total += check(f'3DNA fwd fp16, dilation {dil[0]}, {rows} row(s) per tile, b={B}', lambda: K.sparse3dna_fwd(g, p16, wth))
        L.amdnuwa_set_tuning(16, 0)
        total += check(f'3DNA bwd bf16, dilation {dil[0]}, b={B}', lambda: K.sparse3dna_bwd(g, pbf, wth, dO))
    if only_s3:
        print('TOTAL differing elements:', total)
        return 1 if total else 0
    T = 256
    q = torch.randn(B * n, inner, device=DEV)
    kv = torch.randn(B * T, 2 * inner, device=DEV)
    gx = K.x_geom(B, n, T, heads, dh)
    mask = (torch.rand(B, T, device=DEV) > 0.2).to(torch.uint8)
    nk, nv = torch.randn(heads, dh, device=DEV), torch.randn(heads, dh, device=DEV)
    q16 = K.BF(q.to(torch.bfloat16), None, q.half())
    kv16 = K.BF(kv.to(torch.bfloat16), None, kv.half())
    pk16 = K.xattn_pack(gx, kv16, nk, nv, mask)
    total += check(f'cross attention fwd fp16 (xattn4), b={B}', lambda: K.xattn2_fwd_f16(gx, q16, pk16, wth))
    qb = K.BF(q.to(torch.bfloat16), None)
    pkb = K.xattn_pack(gx, K.BF(kv.to(torch.bfloat16), None), nk, nv, mask)
    total += check(f'cross attention fwd bf16 (xattn4), b={B}', lambda: K.xattn2_fwd(gx, qb, pkb, wth))
    o, stats = K.xattn2_fwd(gx, qb, pkb, wth)
    dO = K.BF(torch.randn(B * n, inner, device=DEV).to(torch.bfloat16), None)
    total += check(f'cross attention bwd (xattn3, query side), b={B}', lambda: (lambda r: (r[0], K.xattn_rows(gx, r[1].hi), K.xattn_rows(gx, r[2].hi), r[3]))(K.xattn2_bwd(gx, qb, dO, pkb, wth, stats)))
    total += other_families(B, check)
    print('TOTAL differing elements:', total)
    return 1 if total else 0


def other_families(B, check):
    """GEMM epilogues (fp16 forward forms, two-MFMA form, GEGLU backward in bf16 and fp16), weight gradients, LayerNorm kernels, cross entropy"""
    total = 0
    torch.manual_seed(1)
    R, D, FP, FFI = B * 2560, 512, 1376, 1365
    x = torch.randn(R, D, device=DEV)
    w, b = torch.randn(D, device=DEV), torch.randn(D, device=DEV)
    h, m1, r1, _ = K.ln_fwd(x, w, b, f16=True)
    w1 = (torch.randn(2 * FP, D, device=DEV) * 0.2).half()
    w2 = (torch.randn(D, FP, device=DEV) * 0.2).half()
    total += check(f'FF1 + gate, fp16 operands (persistent ring), R={R}', lambda: K.gemm_nt_f16ops(h.f16, w1, out_bf16=True, gate=True))
    u, gg16, ggb = K.gemm_nt_f16ops(h.f16, w1, out_bf16=True, gate=True)
    total += check(f'FF2, fp16 operands (K-step 64 ring), R={R}', lambda: K.gemm_nt_f16ops(gg16, w2))
    wqkv = (torch.randn(3 * 512, D, device=DEV) * 0.1).half()
    total += check(f'q/k/v projection, fp16 operands, bf16 + fp16 copies, R={R}', lambda: K.gemm_nt_f16ops(h.f16, wqkv, out_bf16=True, copy_f16=True))
    wo = K.f16_pair(torch.randn(D, 512, device=DEV) * 0.1)
    total += check(f'to_out, two-MFMA form (fp16 x fp16 hi + lo), R={R}', lambda: K.gemm_nt_f16x2(h.f16, wo))
    dy = K.BF((torch.randn(R, D, device=DEV) * 0.01).to(torch.bfloat16), None)
    w2T = K.BF((torch.randn(FP, D, device=DEV) * 0.2).to(torch.bfloat16), None)
    total += check(f'dgrad ff2 + GEGLU backward epilogue, bf16, R={R}', lambda: K.gemm_nt_geglu_bwd(dy, w2T, K.BF(u, None), FP))
    s2 = torch.tensor([1024.0, 1.0 / 1024.0], device=DEV)
    dy16 = (dy.hi.float() * 1024.0).half()
    total += check(f'dgrad ff2 + GEGLU backward epilogue, fp16 gradients, R={R}', lambda: K.gemm_nt_geglu_bwd16(dy16, w2T.hi.half(), u, FP))
    du16 = K.gemm_nt_geglu_bwd16(dy16, w2T.hi.half(), u, FP)
    w1T = w1.t().contiguous()
    total += check(f'dgrad ff1 (long-K four-wave kernel), fp16 gradients, R={R}', lambda: K.gemm_nt_f16ops(du16, w1T, out_f16=True))
    dw1 = torch.empty(2 * FP, D, device=DEV)
    total += check(f'dW ff1 (four-wave TN kernel), fp16, R={R}', lambda: K.gemm_tn16(du16, h.f16, dw1, s2).clone())
    dwb = torch.empty(D, FFI, device=DEV)
    total += check(f'dW ff2 (four-wave TN kernel), bf16, R={R}', lambda: K.gemm_tn(dy, K.BF(ggb, None), dwb, N2=FFI).clone())
    y = torch.randn(R, D, device=DEV)
    total += check(f'ln_post_pre (bf16 + fp16 copies), R={R}', lambda: K.ln_post_pre_fwd(y, x, w, b, b, w, next_shift=(2560, 16), next_f16=True))
    _, m2, r2, _ = K.ln_fwd(y, w, b)
    g = torch.randn(R, D, device=DEV) * 1e-3
    dh = K.BF((torch.randn(R, D, device=DEV) * 1e-3).to(torch.bfloat16), None)
    total += check(f'ln_bwd_chain (bf16 dh), R={R}', lambda: K.ln_bwd_chain(dh, x, m1, r1, w, g, y, m2, r2, w, shift=(2560, 16), want_dsum=True))
    dh16 = K.G16((dh.hi.float() * 1024.0).half(), s2)
    total += check(f'ln_bwd_chain (fp16 dh, fp16 dy_prev), R={R}', lambda: K.ln_bwd_chain(dh16, x, m1, r1, w, g, y, m2, r2, w, shift=(2560, 16), want_dsum=True, out_f16=s2))
    Rc = min(R, 16 * 2560)
    logits = torch.randn(Rc, 8192, device=DEV)
    tg = torch.randint(0, 8192, (Rc,), device=DEV)
    total += check(f'cross entropy (register-resident rows), R={Rc}', lambda: K.ce_fwd(logits, tg, 1.0 / Rc, lo=False))
    return total


if __name__ == '__main__':
    sys.exit(main())
