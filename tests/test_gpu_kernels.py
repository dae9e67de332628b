"""Kernel-level parity on the MI355X: every C-ABI entry point against a plain fp32 restatement
(torch fp32 for the floating-point kernels, oracle/nuwa_oracle.py for the attention cores) on the
same seeded inputs.  Tolerances are max-abs error / max-abs reference and are written next to each
check: operands that are exactly representable in bf16 leave only fp32 accumulation-order noise."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import report, bf_round, to_bf_pair, bf_value  # noqa: E402


@pytest.fixture(scope='module')
def K():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nuwa_pytorch_amd import kernels
    return kernels


@pytest.fixture(scope='module')
def O():
    from oracle import nuwa_oracle
    return nuwa_oracle


DEV = 'cuda'


def shift_ref(x, ntok, fmap):
    from oracle import nuwa_oracle as Or
    B = x.shape[0] // ntok
    return Or.shift_video_tokens(x.reshape(B, ntok, -1).cpu(), fmap).reshape(x.shape)


# ---------------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('M,N,K_', [(128, 128, 64), (200, 72, 96), (1, 8, 8), (257, 513, 40), (2560, 1536, 512),
                                    (96, 8192, 512), (300, 100, 1376)])
@pytest.mark.parametrize('x3', [False, True])
def test_gemm_nt(K, M, N, K_, x3):
    torch.manual_seed(0)
    a = torch.randn(M, K_) * (1 + torch.arange(K_) % 3)       # asymmetric data: catches transposed fragments
    b = torch.randn(N, K_) + 0.25
    if not x3:
        a, b = bf_round(a), bf_round(b)
    A, Bm = to_bf_pair(a.to(DEV), x3), to_bf_pair(b.to(DEV), x3)
    bias = torch.randn(N, device=DEV)
    ref = a.double() @ b.double().t()
    tol = 2e-5 if x3 else 2e-6
    out = K.gemm_nt(A, Bm, bias=bias, alpha=0.5)
    report(f'gemm_nt_f32[{M},{N},{K_},x3={x3}]', out, (0.5 * ref + bias.cpu().double()).float(), tol)
    outb = K.gemm_nt(A, Bm, out_bf16=True)
    report(f'gemm_nt_bf16[{M},{N},{K_},x3={x3}]', outb.hi.float(), ref.float(), 2 ** -8)
    if x3:
        report(f'gemm_nt_bf16hilo[{M},{N},{K_}]', bf_value(outb), ref.float(), 3e-5)


@pytest.mark.parametrize('M,N,K_', [(16384, 2048, 512), (16384 + 37, 2752, 512), (2560 * 8, 1536 + 8, 1376), (256 * 512, 260, 32)])
def test_gemm_nt_x3_256_ring(K, M, N, K_):
    """the bf16x3 256x256 ring (forward GEMMs of the 'bf16x3' / 'bf16x3-fwd' modes) against the first-generation 128x128 x3 kernel
    (tuning key 13 = 1): same accumulation order -> bit-identical; and against fp64 on the hi + lo operand values"""
    from nuwa_pytorch_amd import _lib
    L = _lib.lib()
    torch.manual_seed(1)
    a = (torch.randn(M, K_) * (1 + torch.arange(K_) % 3)).to(DEV)
    b = (torch.randn(N, K_) + 0.25).to(DEV)
    A, Bm = to_bf_pair(a, True), to_bf_pair(b, True)
    bias = torch.randn(N, device=DEV)
    ref = (bf_value(A).double() @ bf_value(Bm).double().t()) * 0.5 + bias.double()
    try:
        L.amdnuwa_set_tuning(13, 1)
        old_f = K.gemm_nt(A, Bm, bias=bias, alpha=0.5)
        old_b = K.gemm_nt(A, Bm, bias=bias, alpha=0.5, out_bf16=True)
    finally:
        L.amdnuwa_set_tuning(13, 0)
    new_f = K.gemm_nt(A, Bm, bias=bias, alpha=0.5)
    new_b = K.gemm_nt(A, Bm, bias=bias, alpha=0.5, out_bf16=True)
    report(f'gemm_nt_x3_256[{M},{N},{K_}] f32 vs fp64', new_f, ref.float(), 2e-5)
    report(f'gemm_nt_x3_256[{M},{N},{K_}] hi+lo vs fp64', bf_value(new_b), ref.float(), 3e-5)
    assert torch.equal(new_f, old_f), f'fp32 output differs from the 128x128 x3 kernel: {(new_f - old_f).abs().max().item():.3e}'
    assert torch.equal(new_b.hi, old_b.hi) and torch.equal(new_b.lo, old_b.lo)


@pytest.mark.parametrize('B,ntok,fmap,D,N', [(2, 17, 4, 32, 24), (3, 48, 4, 64, 130), (2, 129, 8, 128, 64), (1, 2561, 16, 512, 256)])
def test_gemm_nt_shift_loader(K, B, ntok, fmap, D, N):
    torch.manual_seed(1)
    a = bf_round(torch.randn(B * ntok, D))
    w = bf_round(torch.randn(N, D))
    ref = shift_ref(a, ntok, fmap).double() @ w.double().t()
    out = K.gemm_nt(to_bf_pair(a.to(DEV), False), to_bf_pair(w.to(DEV), False), shift=(ntok, fmap))
    report(f'gemm_nt_shift[{B},{ntok},{fmap},{D}]', out, ref.float(), 2e-6)


@pytest.mark.parametrize('R,N1,N2', [(64, 128, 128), (300, 72, 40), (2560, 1536, 512), (5000, 85, 32), (33, 8, 8), (4096, 1365, 512)])
@pytest.mark.parametrize('x3', [False, True])
def test_gemm_tn(K, R, N1, N2, x3):
    torch.manual_seed(2)
    ld1, ld2 = (N1 + 7) // 8 * 8, (N2 + 7) // 8 * 8
    a = torch.randn(R, ld1) * (1 + torch.arange(ld1) % 5)
    b = torch.randn(R, ld2) - 0.3
    if not x3:
        a, b = bf_round(a), bf_round(b)
    A, Bm = to_bf_pair(a.to(DEV), x3), to_bf_pair(b.to(DEV), x3)
    from nuwa_pytorch_amd.kernels import BF, view
    out = torch.full((N1, N2), 7.0, device=DEV)
    K.gemm_tn(view(A, cols=slice(0, N1)), view(Bm, cols=slice(0, N2)), out, alpha=2.0, beta=0.0, N1=N1, N2=N2)
    ref = 2.0 * (a[:, :N1].double().t() @ b[:, :N2].double())
    report(f'gemm_tn[{R},{N1},{N2},x3={x3}]', out, ref.float(), 3e-5 if x3 else 3e-6)
    out2 = torch.ones((N1, N2), device=DEV)
    K.gemm_tn(view(A, cols=slice(0, N1)), view(Bm, cols=slice(0, N2)), out2, alpha=1.0, beta=1.0, N1=N1, N2=N2)
    report(f'gemm_tn_beta[{R},{N1},{N2},x3={x3}]', out2, (ref / 2 + 1).float(), 3e-5 if x3 else 3e-6)


@pytest.mark.parametrize('B,ntok,fmap,D,N1', [(2, 17, 4, 32, 24), (2, 129, 8, 128, 200), (1, 2561, 16, 512, 64)])
def test_gemm_tn_shift_loader(K, B, ntok, fmap, D, N1):
    torch.manual_seed(3)
    dy = bf_round(torch.randn(B * ntok, (N1 + 7) // 8 * 8))
    h = bf_round(torch.randn(B * ntok, D))
    from nuwa_pytorch_amd.kernels import view
    out = torch.empty((N1, D), device=DEV)
    K.gemm_tn(view(to_bf_pair(dy.to(DEV), False), cols=slice(0, N1)), to_bf_pair(h.to(DEV), False), out, shift=(ntok, fmap), N1=N1)
    ref = dy[:, :N1].double().t() @ shift_ref(h, ntok, fmap).double()
    report(f'gemm_tn_shift[{B},{ntok},{fmap},{D}]', out, ref.float(), 3e-6)


# ---------------------------------------------------------------------------------------------------
# row kernels
# ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('R,D', [(7, 32), (130, 512), (64, 1024), (33, 48)])
def test_layernorm_fwd_bwd(K, R, D):
    torch.manual_seed(4)
    x = (torch.randn(R, D) * 2 + 0.5).requires_grad_(True)
    res = torch.randn(R, D, requires_grad=True)
    w, b = torch.randn(D, requires_grad=True), torch.randn(D, requires_grad=True)
    y_ref = F.layer_norm(x, (D,), w, b)
    g = torch.randn(R, D)
    (y_ref + res).backward(g)
    xd, rd, wd, bd = (t.detach().to(DEV) for t in (x, res, w, b))
    K.set_precision('bf16x3')
    try:
        out, m, r, _ = K.ln_fwd(xd, wd, bd)
        report(f'ln_fwd_pre[{R},{D}]', bf_value(out), y_ref.detach(), 2e-5)
        yo, m2, r2 = K.ln_fwd(xd, wd, bd, resid=rd)
        report(f'ln_fwd_post[{R},{D}]', yo, (y_ref + res).detach(), 2e-6)
        # resid - LN(x) (the reversible reconstruction x2 = y2 - g(y1)): the bits of -((-resid) + LN(x)), the two-negation form it replaced
        ym, _, _ = K.ln_fwd(xd, wd, bd, resid=rd, minus=True)
        yn, _, _ = K.ln_fwd(xd, wd, bd, resid=-rd)
        assert torch.equal(ym, -yn)
        report(f'ln_fwd_post_minus[{R},{D}]', ym, (res - y_ref).detach(), 2e-6)
        dx, dw, db, ds = K.ln_bwd(g.to(DEV), xd, m2, r2, wd, to_bf=True, want_dsum=True)
        report(f'ln_bwd_dx_bf[{R},{D}]', bf_value(dx), x.grad, 3e-5)
        report(f'ln_bwd_dw[{R},{D}]', dw, w.grad, 1e-5)
        report(f'ln_bwd_db[{R},{D}]', db, b.grad, 1e-5)
        report(f'ln_bwd_dsum[{R},{D}]', ds, x.grad.sum(0), 1e-4)
        dres = torch.randn(R, D, device=DEV)
        dx2, _, _, _ = K.ln_bwd(g.to(DEV), xd, m2, r2, wd, dres=dres)
        report(f'ln_bwd_dx_acc[{R},{D}]', dx2, x.grad + dres.cpu(), 1e-5)
    finally:
        K.set_precision('bf16')


@pytest.mark.parametrize('R,D', [(130, 512), (33, 48)])
def test_layernorm_bf16_inputs(K, R, D):
    """fast-mode forms: the LN input (GEMM output) or the incoming gradient arrive as bf16; results must equal the fp32
    kernels run on the same bf16-rounded values"""
    torch.manual_seed(14)
    xb = (torch.randn(R, D) * 2 + 0.5).to(torch.bfloat16)
    gb = torch.randn(R, D).to(torch.bfloat16)
    x32, g32 = xb.float().to(DEV), gb.float().to(DEV)
    res, g = torch.randn(R, D, device=DEV), torch.randn(R, D, device=DEV)
    w, b = torch.randn(D, device=DEV), torch.randn(D, device=DEV)
    xbf, gbf = K.BF(xb.to(DEV), None), K.BF(gb.to(DEV), None)
    y0, m0, r0 = K.ln_fwd(x32, w, b, resid=res)
    y1, m1, r1 = K.ln_fwd(xbf, w, b, resid=res)
    assert torch.equal(y0, y1) and torch.equal(m0, m1) and torch.equal(r0, r1)
    a0 = K.ln_bwd(g, x32, m0, r0, w, to_bf=True, want_dsum=True)
    a1 = K.ln_bwd(g, xbf, m0, r0, w, to_bf=True, want_dsum=True)
    assert torch.equal(a0[0].hi, a1[0].hi) and all(torch.equal(p, q) for p, q in zip(a0[1:], a1[1:]))
    for shift in (None, (11, 2)) if R == 33 else (None,):
        b0 = K.ln_bwd(g32, res, m0, r0, w, dres=g, shift=shift)
        b1 = K.ln_bwd(gbf, res, m0, r0, w, dres=g, shift=shift)
        assert all(torch.equal(p, q) for p, q in zip(b0[:3], b1[:3]))


@pytest.mark.parametrize('B,ntok,fmap,D', [(2, 23, 4, 32), (1, 49, 4, 64), (2, 2, 4, 32), (1, 2561, 16, 512)])
def test_layernorm_fwd_with_folded_token_shift(K, O, B, ntok, fmap, D):
    """pre-LN writing shift(LN(x)) directly (partial last frame, n = 2, full cfg-3 row count)"""
    torch.manual_seed(6)
    x = torch.randn(B, ntok, D) * 1.5 + 0.2
    w, b = torch.randn(D), torch.randn(D)
    ref = O.shift_video_tokens(F.layer_norm(x, (D,), w, b), fmap)
    K.set_precision('bf16x3')
    try:
        out, m, r, _ = K.ln_fwd(x.reshape(B * ntok, D).to(DEV), w.to(DEV), b.to(DEV), shift=(ntok, fmap))
        report(f'ln_fwd_shift[{B},{ntok},{fmap},{D}]', bf_value(out).reshape(B, ntok, D), ref, 2e-5)
    finally:
        K.set_precision('bf16')


@pytest.mark.parametrize('B,ntok,fmap,D,ybf', [(2, 23, 4, 32, False), (3, 17, 4, 512, True), (2, 10, None, 1024, True),
                                                  (1, 5, None, 64, False)])
def test_layernorm_post_pre_chain(K, O, B, ntok, fmap, D, ybf):
    """post-norm + residual of block k fused with the pre-norm (+ shift) of block k+1: identical, bit for bit, to the two
    separate kernels, and equal to the oracle's two layer norms"""
    torch.manual_seed(8)
    R = B * ntok
    y = (torch.randn(R, D) * 1.3).to(DEV)
    resid = torch.randn(R, D).to(DEV)
    w, b, w2, b2 = (torch.randn(D).to(DEV) for _ in range(4))
    shift = (ntok, fmap) if fmap else None
    yin = y
    if ybf:
        hi = K.empty_bf((R, D), DEV)
        K.cast_pad(y, hi)
        yin = K.BF(hi.hi, None)
    xo_a, m_a, r_a = K.ln_fwd(yin, w, b, resid=resid)
    h_a, m1_a, r1_a, _ = K.ln_fwd(xo_a, w2, b2, shift=shift)
    xo, m, r, h, m1, r1 = K.ln_post_pre_fwd(yin, resid, w, b, w2, b2, next_shift=shift)
    for name, u, v in (('xo', xo, xo_a), ('m', m, m_a), ('r', r, r_a), ('h', h.hi, h_a.hi), ('m1', m1, m1_a), ('r1', r1, r1_a)):
        assert torch.equal(u, v), name
    yv = bf_value(yin).cpu() if ybf else y.cpu()
    x_ref = resid.cpu() + F.layer_norm(yv, (D,), w.cpu(), b.cpu())
    h_ref = F.layer_norm(x_ref, (D,), w2.cpu(), b2.cpu()).reshape(B, ntok, D)
    if fmap:
        h_ref = O.shift_video_tokens(h_ref, fmap)
    report(f'ln_post_pre.x[{R},{D}]', xo, x_ref, 2e-5)
    report(f'ln_post_pre.h[{R},{D}]', bf_value(h).reshape(B, ntok, D), h_ref, 8e-3)


@pytest.mark.parametrize('B,ntok,fmap,D,bf', [(2, 23, 4, 32, False), (3, 17, 4, 512, True), (2, 10, None, 1024, True),
                                                 (1, 5, None, 64, False), (3, 17, 4, 512, 'dh'), (2, 23, 4, 32, 'dh')])
def test_layernorm_bwd_chain(K, O, B, ntok, fmap, D, bf):
    """pre-norm backward of block k+1 and post-norm backward of block k in one pass: same dx / dy_prev as the two separate
    kernels (bit for bit), weight gradients equal up to the order of the per-workgroup partial sums, all equal to autograd.
    bf: False = fp32 dh and y_prev, True = both bf16, 'dh' = bf16 dh with an fp32 y_prev (the 'bf16x3-fwd' mode's form)"""
    torch.manual_seed(12)
    R = B * ntok
    shift = (ntok, fmap) if fmap else None
    x = torch.randn(R, D) * 1.3 + 0.1
    yprev = torch.randn(R, D)
    dh = torch.randn(R, D)
    g = torch.randn(R, D)
    if bf:
        dh = dh.bfloat16().float()
    if bf is True:
        yprev = yprev.bfloat16().float()
    w, w_prev = torch.randn(D), torch.randn(D)
    # autograd reference:  x is the stream row; h = shift(LN(x; w)) receives dh; the row also receives g directly.
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    h = F.layer_norm(xr, (D,), wr, torch.zeros(D)).reshape(B, ntok, D)
    if fmap:
        h = O.shift_video_tokens(h, fmap)
    (h.reshape(R, D) * dh).sum().backward()
    dx_ref = g + xr.grad
    yr = yprev.clone().requires_grad_(True)
    wpr = w_prev.clone().requires_grad_(True)
    bpr = torch.zeros(D, requires_grad=True)
    (F.layer_norm(yr, (D,), wpr, bpr) * dx_ref).sum().backward()
    dev = lambda t: t.to(DEV)
    xd = dev(x)
    _, m1, r1, _ = K.ln_fwd(xd, dev(w), dev(torch.zeros(D)))
    zero = torch.zeros(R, D, device=DEV)
    _, m2, r2 = K.ln_fwd(dev(yprev), dev(w_prev), dev(torch.zeros(D)), resid=zero)
    K.set_precision('bf16' if bf else 'bf16x3')
    try:
        def form(t, as_bf):
            return K.BF(dev(t).bfloat16(), None) if as_bf else dev(t)
        fdh, fy = form(dh, bool(bf)), form(yprev, bf is True)
        dx, dw, db, dyp, dwp, dbp, dsp = K.ln_bwd_chain(fdh, xd, m1, r1, dev(w), dev(g), fy, m2, r2, dev(w_prev),
                                                        shift=shift, want_dsum=True)
        dx_a, dw_a, db_a, _ = K.ln_bwd(fdh, xd, m1, r1, dev(w), dres=dev(g), shift=shift)
        dy_a, dwp_a, dbp_a, dsp_a = K.ln_bwd(dx_a, fy, m2, r2, dev(w_prev), to_bf=True, want_dsum=True)
    finally:
        K.set_precision('bf16')
    assert torch.equal(dx, dx_a) and torch.equal(dyp.hi, dy_a.hi)
    for nm, u, v in (('dw', dw, dw_a), ('db', db, db_a), ('dwp', dwp, dwp_a), ('dbp', dbp, dbp_a), ('dsp', dsp, dsp_a)):
        report(f'ln_bwd_chain.{nm}[{R},{D}]', u, v, 2e-6)
    report(f'ln_bwd_chain.dx[{R},{D}]', dx, dx_ref, 2e-5)
    report(f'ln_bwd_chain.dy_prev[{R},{D}]', bf_value(dyp), yr.grad, 2 ** -7 if bf else 3e-5)
    report(f'ln_bwd_chain.dw[{R},{D}]', dw, wr.grad, 2e-5)
    report(f'ln_bwd_chain.dw_prev[{R},{D}]', dwp, wpr.grad, 2e-5)
    report(f'ln_bwd_chain.db_prev[{R},{D}]', dbp, bpr.grad, 2e-5)


def test_layernorm_bwd_inverse_shift(K, O):
    torch.manual_seed(5)
    B, ntok, fmap, D = 2, 23, 4, 32
    x = torch.randn(B, ntok, D, requires_grad=True)
    w, b = torch.randn(D), torch.randn(D)
    hs = O.shift_video_tokens(O.layer_norm(x, w, b), fmap)
    g = torch.randn(B, ntok, D)
    hs.backward(g)
    xd = x.detach().reshape(B * ntok, D).to(DEV)
    _, m, r, _ = K.ln_fwd(xd, w.to(DEV), b.to(DEV))
    dx, dw, db, _ = K.ln_bwd(g.reshape(B * ntok, D).to(DEV), xd, m, r, w.to(DEV), shift=(ntok, fmap))
    report('ln_bwd_inverse_shift', dx, x.grad.reshape(B * ntok, D), 1e-5)


def test_stable_layernorm(K, O):
    torch.manual_seed(6)
    R, D = 50, 64
    x = torch.randn(R, D, requires_grad=True)
    w, b = torch.randn(D, requires_grad=True), torch.randn(D, requires_grad=True)
    y = O.stable_layer_norm(x, w, b)
    g = torch.randn(R, D)
    y.backward(g)
    K.set_precision('bf16x3')
    try:
        xd = x.detach().to(DEV)
        out, m, r, ia = K.ln_fwd(xd, w.detach().to(DEV), b.detach().to(DEV), stable=True)
        report('stable_ln_fwd', bf_value(out), y.detach(), 2e-5)
        dx, dw, db, _ = K.ln_bwd(g.to(DEV), xd, m, r, w.detach().to(DEV), inv_amax=ia)
        report('stable_ln_dx', dx, x.grad, 2e-5)
        report('stable_ln_dw', dw, w.grad, 1e-5)
    finally:
        K.set_precision('bf16')


def test_geglu(K):
    torch.manual_seed(7)
    R, FP = 37, 96
    u = bf_round(torch.randn(R, 2 * FP)).requires_grad_(True)
    a, g_ = u[:, :FP], u[:, FP:]
    y = a * F.gelu(g_)
    d = bf_round(torch.randn(R, FP))
    y.backward(d)
    ub = to_bf_pair(u.detach().to(DEV), False)
    K.set_precision('bf16x3')
    try:
        from nuwa_pytorch_amd.kernels import BF
        ub = BF(ub.hi, torch.zeros_like(ub.hi))
        o = K.geglu_fwd(ub, FP)
        report('geglu_fwd', bf_value(o), y.detach(), 2e-5)
        db = to_bf_pair(d.to(DEV), True)
        du = K.geglu_bwd(ub, db, FP)
        report('geglu_bwd', bf_value(du), u.grad, 3e-5)
    finally:
        K.set_precision('bf16')


@pytest.mark.parametrize('x3', [False, True])
def test_geglu_interleaved_layout(K, x3):
    """the interleaved-by-8 form of u (8 values, their 8 gates, ...) that the FF1 GEMM epilogue produces: same numbers as the
    [a | gate] form, and interleave / de-interleave are inverse permutations"""
    torch.manual_seed(15)
    R, FP = 37, 96
    u = torch.randn(R, 2 * FP)
    dy = torch.randn(R, FP)
    if not x3:
        u, dy = bf_round(u), bf_round(dy)
    assert torch.equal(K.geglu_deinterleave(K.geglu_interleave(u, FP, dim=1), FP, dim=1), u)
    ub, db = to_bf_pair(u.to(DEV), x3), to_bf_pair(dy.to(DEV), x3)
    uil = to_bf_pair(K.geglu_interleave(u, FP, dim=1).contiguous().to(DEV), x3)
    o_ref, o_il = K.geglu_fwd(ub, FP), K.geglu_fwd(uil, FP, interleaved=True)
    assert torch.equal(o_ref.hi, o_il.hi)
    du_ref, du_il = K.geglu_bwd(ub, db, FP), K.geglu_bwd(uil, db, FP, interleaved=True)
    assert torch.equal(K.geglu_deinterleave(du_il.hi, FP, dim=1), du_ref.hi)
    if x3:
        assert torch.equal(o_ref.lo, o_il.lo) and torch.equal(K.geglu_deinterleave(du_il.lo, FP, dim=1), du_ref.lo)


@pytest.mark.parametrize('M,N,Kd,x3', [(16384, 2752, 512, False), (300, 96, 64, False), (300, 96, 64, True), (8, 2752, 512, False)])
def test_gemm_nt_with_geglu_epilogue(K, M, N, Kd, x3):
    """FF1: u = h W1^T and a * gelu(gate) from one call.  On the 256x256 ring (the first shape) the gate runs in the GEMM
    epilogue; elsewhere the library falls back to GEMM + gate kernel.  Both must equal the two separate calls bit for bit."""
    torch.manual_seed(M % 97)
    a = torch.randn(M, Kd) * 0.5
    w = torch.randn(N, Kd) * 0.2
    if not x3:
        a, w = bf_round(a), bf_round(w)
    ap, wp = to_bf_pair(a.to(DEV), x3), to_bf_pair(w.to(DEV), x3)
    FP = N // 2
    K.set_precision('bf16x3' if x3 else 'bf16')
    try:
        u_ref = K.gemm_nt(ap, wp, out_bf16=True)
        gg_ref = K.geglu_fwd(u_ref, FP, interleaved=True)
        gg = K.empty_bf((M, FP), DEV)
        u = K.gemm_nt(ap, wp, out_bf16=True, geglu_out=gg)
    finally:
        K.set_precision('bf16')
    assert torch.equal(u.hi, u_ref.hi) and torch.equal(gg.hi, gg_ref.hi)
    if x3:
        assert torch.equal(gg.lo, gg_ref.lo)
    ud = K.geglu_deinterleave(bf_value(u).cpu(), FP, dim=1)
    ref = (a.double() @ K.geglu_deinterleave(w, FP, dim=0).double().t()).float()
    report(f'gemm_geglu.u[{M},{N},x3={x3}]', ud, ref, 2e-5 if x3 else 2 ** -8)
    y_ref = ud[:, :FP] * F.gelu(ud[:, FP:])
    report(f'gemm_geglu.gg[{M},{N},x3={x3}]', bf_value(gg).cpu(), y_ref, 2e-5 if x3 else 2 ** -7)


@pytest.mark.parametrize('mode', ['bf16x3', 'bf16x3-fwd'])
def test_gemm_nt_x3_ring_with_geglu_epilogue(K, mode):
    """FF1 on the bf16x3 256x256 ring: the gate runs on the fp32 accumulators in the epilogue (hi + lo gate output).  Against the
    two separate calls (gate recomputed from the stored u hi + lo: agrees to the pair's 2^-17) and against fp64; in 'bf16x3-fwd'
    u's lo part is not written at all."""
    M, N, Kd = 16384, 2752, 512
    torch.manual_seed(5)
    a = torch.randn(M, Kd) * 0.5
    w = torch.randn(N, Kd) * 0.2
    ap, wp = to_bf_pair(a.to(DEV), True), to_bf_pair(w.to(DEV), True)
    FP = N // 2
    K.set_precision(mode)
    try:
        u_ref = K.gemm_nt(ap, wp, out_bf16=True)
        gg_ref = K.geglu_fwd(u_ref, FP, interleaved=True)
        gg = K.empty_bf((M, FP), DEV)
        u = K.gemm_nt(ap, wp, out_bf16=True, geglu_out=gg)
    finally:
        K.set_precision('bf16')
    assert torch.equal(u.hi, u_ref.hi)
    assert (u.lo is None) == (mode == 'bf16x3-fwd')
    if u.lo is not None:
        assert torch.equal(u.lo, u_ref.lo)
    assert gg.lo is not None
    report(f'gemm_x3_geglu.gg_vs_separate[{mode}]', bf_value(gg), bf_value(gg_ref), 2e-5)
    ud = K.geglu_deinterleave((bf_value(ap).double() @ bf_value(wp).double().t()).float().cpu(), FP, dim=1)
    y_ref = ud[:, :FP] * F.gelu(ud[:, FP:])
    report(f'gemm_x3_geglu.gg_vs_fp64[{mode}]', bf_value(gg).cpu(), y_ref, 3e-5)


@pytest.mark.parametrize('M,FP,Kd,x3', [(16384, 1376, 512, False), (300, 48, 64, False), (300, 48, 64, True), (8, 1376, 512, False)])
def test_gemm_nt_with_geglu_backward_epilogue(K, M, FP, Kd, x3):
    """FF backward through the gate: dgg = dy W2 and du from one call (in the GEMM epilogue on the 256x256 ring, dgg never
    written) must equal GEMM + stand-alone gate kernel bit for bit, and autograd through a * gelu(gate)"""
    torch.manual_seed(FP % 89)
    dy = torch.randn(M, Kd) * 0.5
    w2T = torch.randn(FP, Kd) * 0.2                    # dgg = dy @ w2T^T
    u = torch.randn(M, 2 * FP)                         # interleaved layout
    if not x3:
        dy, w2T, u = bf_round(dy), bf_round(w2T), bf_round(u)
    dyp, wp, up = to_bf_pair(dy.to(DEV), x3), to_bf_pair(w2T.to(DEV), x3), to_bf_pair(u.to(DEV), x3)
    K.set_precision('bf16x3' if x3 else 'bf16')
    try:
        dgg_ref = K.gemm_nt(dyp, wp, out_bf16=True)
        du_ref = K.geglu_bwd(up, dgg_ref, FP, interleaved=True)
        du = K.gemm_nt_geglu_bwd(dyp, wp, up, FP)
    finally:
        K.set_precision('bf16')
    assert torch.equal(du.hi, du_ref.hi)
    if x3:
        assert torch.equal(du.lo, du_ref.lo)
    ud = K.geglu_deinterleave(u, FP, dim=1).requires_grad_(True)
    y = ud[:, :FP] * F.gelu(ud[:, FP:])
    y.backward(bf_value(dgg_ref).cpu())
    report(f'gemm_geglu_bwd.du[{M},{FP},x3={x3}]', K.geglu_deinterleave(bf_value(du).cpu(), FP, dim=1), ud.grad, 3e-5 if x3 else 2 ** -7)


def test_casts(K):
    torch.manual_seed(8)
    w = torch.randn(70, 52, device=DEV)
    K.set_precision('bf16x3')
    try:
        out = K.zeros_bf((80, 64), DEV)
        K.cast_pad(w, out, row0=5, Cp=64)
        v = bf_value(out)
        report('cast_pad', v[5:75, :52], w.cpu(), 2e-5)
        assert float(v[:5].abs().max()) == 0 and float(v[5:75, 52:].abs().max()) == 0
        outT = K.zeros_bf((52, 90), DEV)
        K.transpose_cast(w, outT, col0=10)
        report('transpose_cast', bf_value(outT)[:, 10:80], w.t().cpu(), 2e-5)
    finally:
        K.set_precision('bf16')


def test_embed_fwd_bwd(K, O):
    torch.manual_seed(9)
    B, Fr, H, W, D, C = 2, 3, 4, 4, 32, 20
    N = Fr * H * W
    P = {'image_embedding.embed.weight': torch.randn(C, D, requires_grad=True), 'video_bos': torch.randn(D, requires_grad=True),
         'video_pos_emb.axial1': torch.randn(Fr, D, requires_grad=True), 'video_pos_emb.axial2': torch.randn(H, D, requires_grad=True),
         'video_pos_emb.axial3': torch.randn(W, D, requires_grad=True)}
    for n1 in (N - 1, 21, 1):
        ids = torch.randint(0, C, (B, n1))
        for p in P.values():
            p.grad = None
        x = O.embed_assemble(ids, P, training=True, frac=0.2)
        g = torch.randn_like(x)
        x.backward(g)
        d = {k: v.detach().to(DEV) for k, v in P.items()}
        xk = K.embed_fwd(ids.to(DEV), d['image_embedding.embed.weight'], d['video_pos_emb.axial1'], d['video_pos_emb.axial2'],
                         d['video_pos_emb.axial3'], d['video_bos'], B, n1 + 1, H, W, 0.2)
        report(f'embed_fwd[n1={n1}]', xk.reshape(B, n1 + 1, D), x.detach(), 1e-6)
        dW = torch.zeros(C, D, device=DEV)
        d1, d2, d3 = (torch.zeros(s, D, device=DEV) for s in (Fr, H, W))
        db = torch.zeros(D, device=DEV)
        K.embed_bwd(ids.to(DEV), g.reshape(B * (n1 + 1), D).to(DEV), dW, d1, d2, d3, db, B, n1 + 1, Fr, H, W, 0.2)
        report(f'embed_bwd_dW[n1={n1}]', dW, P['image_embedding.embed.weight'].grad, 1e-5)
        report(f'embed_bwd_ax1[n1={n1}]', d1, P['video_pos_emb.axial1'].grad, 1e-5)
        report(f'embed_bwd_ax2[n1={n1}]', d2, P['video_pos_emb.axial2'].grad, 1e-5)
        report(f'embed_bwd_ax3[n1={n1}]', d3, P['video_pos_emb.axial3'].grad, 1e-5)
        report(f'embed_bwd_bos[n1={n1}]', db, P['video_bos'].grad, 1e-5)


@pytest.mark.parametrize('case', ['one code', 'few codes', 'uniform', 'alternating blocks'])
def test_embed_bwd_long_runs(K, case):
    """the token-embedding gradient on skewed id distributions (raw frames through the tokenizer land on a few codes): runs of equal ids
    that span many of the kernel's segments, against an fp64 index_add; bit-reproducible; and not serial in the run length"""
    torch.manual_seed(5)
    B, ntok, D, C = 8, 2560, 512, 8192
    n1 = ntok - 1
    if case == 'one code':
        ids = torch.full((B, n1), 77)
    elif case == 'few codes':
        ids = torch.tensor([3, 4000, 8191])[torch.randint(0, 3, (B, n1))]
        ids[0, :100] = torch.randint(0, C, (100,))
    elif case == 'uniform':
        ids = torch.randint(0, C, (B, n1))
    else:
        ids = (torch.arange(B * n1) // 700).reshape(B, n1) * 13 % C
    dx = torch.randn(B * ntok, D)
    ref = torch.zeros(C, D, dtype=torch.float64)
    rows = (torch.arange(B)[:, None] * ntok + 1 + torch.arange(n1)[None, :]).reshape(-1)
    ref.index_add_(0, ids.reshape(-1), 0.2 * dx[rows].double())
    outs = []
    for _ in range(2):
        dW = torch.zeros(C, D, device=DEV)
        d1, d2, d3 = (torch.zeros(s_, D, device=DEV) for s_ in (10, 16, 16))
        db = torch.zeros(D, device=DEV)
        K.embed_bwd(ids.to(DEV), dx.to(DEV), dW, d1, d2, d3, db, B, ntok, 10, 16, 16, 0.2)
        outs.append(dW)
    assert torch.equal(outs[0], outs[1])
    report(f'embed_bwd_long_runs[{case}]', outs[0], ref.float(), 2e-6)
    # device time of the call alone (inputs resident, best of three: no host hiccup or copy inside the bound)
    ids_d, dx_d = ids.to(DEV), dx.to(DEV)
    best = float('inf')
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.embed_bwd(ids_d, dx_d, outs[0], d1, d2, d3, db, B, ntok, 10, 16, 16, 0.2)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    assert best < 30.0, f'a run of equal ids must not be walked by one wave ({best:.1f} ms)'


@pytest.mark.parametrize('R,C', [(5, 64), (300, 8192), (17, 1000)])
def test_cross_entropy(K, R, C):
    torch.manual_seed(10)
    logits = (torch.randn(R, C) * 3).requires_grad_(True)
    t = torch.randint(0, C, (R,))
    loss = F.cross_entropy(logits, t)
    loss.backward()
    K.set_precision('bf16x3')
    try:
        lk, dl = K.ce_fwd(logits.detach().to(DEV), t.to(DEV), 1.0 / R)
        report(f'ce_loss[{R},{C}]', lk.reshape(1), loss.detach().reshape(1), 1e-6)
        report(f'ce_dlogits[{R},{C}]', bf_value(dl), logits.grad, 3e-5)
    finally:
        K.set_precision('bf16')


# ---------------------------------------------------------------------------------------------------
# attention cores vs the oracle
# ---------------------------------------------------------------------------------------------------

S3_CASES = [((3, 4, 4), (3, 3, 3), (1, 1, 1), 2, 32, None), ((3, 4, 4), (3, 3, 3), (2, 2, 2), 2, 32, None),
            ((4, 8, 8), (5, 3, 3), (1, 1, 1), 4, 64, None), ((4, 8, 8), (5, 3, 3), (2, 2, 2), 8, 64, None),
            ((4, 8, 8), (5, 3, 3), (4, 4, 4), 8, 32, None), ((3, 4, 4), (3, 3, 3), (1, 1, 1), 2, 32, 2),
            ((3, 4, 4), (3, 3, 3), (1, 1, 1), 2, 32, 17), ((3, 4, 4), (3, 3, 3), (1, 2, 1), 3, 64, 23),
            ((2, 16, 16), (5, 3, 3), (1, 1, 1), 8, 64, None), ((3, 16, 16), (3, 3, 3), (4, 4, 4), 8, 64, 300)]


@pytest.mark.parametrize('case', range(len(S3_CASES)))
@pytest.mark.parametrize('x3', [False, True])
def test_sparse3dna_core(K, O, case, x3):
    shape, kern, dil, heads, dh, n = S3_CASES[case]
    N = shape[0] * shape[1] * shape[2]
    n = N if n is None else n
    B = 2
    inner = heads * dh
    torch.manual_seed(11 + case)
    qkv = torch.randn(B, n, 3, heads, dh)
    if not x3:
        qkv = bf_round(qkv)
    qkv.requires_grad_(True)
    wth = torch.randn(heads, heads) * 0.5 + torch.eye(heads)
    wth.requires_grad_(True)
    idx = O.neighbor_table(shape, kern, dil, causal=True)
    o_ref = O.sparse3dna_core(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], wth, idx, dh ** -0.5)
    do = torch.randn_like(o_ref)
    if not x3:
        do = bf_round(do)
    o_ref.backward(do)
    g = K.s3_geom(B, n, shape, kern, dil, heads, dh)
    qkvp = to_bf_pair(qkv.detach().reshape(B * n, 3 * inner).to(DEV), x3)
    o = K.sparse3dna_fwd(g, qkvp, wth.detach().to(DEV))
    # bf16 mode: the output is rounded to bf16 (2^-9) and the post-softmax probability of each tap enters the
    # packed v_dot2 accumulation as a bf16 coefficient (another 2^-9 per term); parity mode is all-fp32
    tol_o = 3e-5 if x3 else 2 ** -7
    tag = f'[{case},x3={x3}]'
    report('s3_fwd' + tag, (bf_value(o) if x3 else o.hi.float()).reshape(B, n, heads, dh), o_ref.detach(), tol_o)
    dqkv, dwth, _ = K.sparse3dna_bwd(g, qkvp, wth.detach().to(DEV), to_bf_pair(do.reshape(B * n, inner).to(DEV), x3))
    gq = qkv.grad.reshape(B * n, 3 * inner)
    got = bf_value(dqkv) if x3 else dqkv.hi.float()
    tol_g = 5e-5 if x3 else 2 ** -6
    for nm, sl in (('dq', slice(0, inner)), ('dk', slice(inner, 2 * inner)), ('dv', slice(2 * inner, 3 * inner))):
        if n > 1 or nm == 'dv':
            report(f's3_bwd_{nm}' + tag, got[:, sl], gq[:, sl], tol_g)
    if n > 1:
        report('s3_bwd_dwth' + tag, dwth, wth.grad, 1e-4)


@pytest.mark.parametrize('shape,kern,dil,n', [((2, 16, 16), (3, 3, 3), (1, 2, 1), None), ((3, 16, 16), (5, 3, 3), (2, 1, 4), 530)])
def test_sparse3dna_core_rel_pos_bias_on_the_mfma_kernels(K, O, shape, kern, dil, n):
    """W = 16, 8 heads x 64 (the geometry the MFMA forward / backward kernels take) with per-axis dilations, a partial last
    frame and the relative-position bias: output, dq / dk / dv, dW_th and d(bias) against autograd through the oracle"""
    heads, dh, B = 8, 64, 2
    N = shape[0] * shape[1] * shape[2]
    n = N if n is None else n
    inner = heads * dh
    J = kern[0] * kern[1] * kern[2] + 1
    torch.manual_seed(23)
    qkv = bf_round(torch.randn(B, n, 3, heads, dh)).requires_grad_(True)
    wth = (torch.randn(heads, heads) * 0.5 + torch.eye(heads)).requires_grad_(True)
    rel = (torch.randn(heads, J - 1) * 0.7).requires_grad_(True)                   # oracle layout (h, K)
    idx = O.neighbor_table(shape, kern, dil, causal=True)
    o_ref = O.sparse3dna_core(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], wth, idx, dh ** -0.5, rel_pos_bias=rel)
    do = bf_round(torch.randn_like(o_ref))
    o_ref.backward(do)
    g = K.s3_geom(B, n, shape, kern, dil, heads, dh)
    qkvp = to_bf_pair(qkv.detach().reshape(B * n, 3 * inner).to(DEV), False)
    rel_dev = torch.cat((torch.zeros(1, heads), rel.detach().t()), 0).contiguous().to(DEV)   # kernel layout [J, heads], slot 0 = <bos>
    o = K.sparse3dna_fwd(g, qkvp, wth.detach().to(DEV), rel_bias=rel_dev)
    report('s3_mfma_rel.fwd', o.hi.float().reshape(B, n, heads, dh), o_ref.detach(), 2 ** -7)
    dqkv, dwth, drel = K.sparse3dna_bwd(g, qkvp, wth.detach().to(DEV), to_bf_pair(do.reshape(B * n, inner).to(DEV), False),
                                        rel_bias=rel_dev)
    gq = qkv.grad.reshape(B * n, 3 * inner)
    got = dqkv.hi.float()
    for nm, sl in (('dq', slice(0, inner)), ('dk', slice(inner, 2 * inner)), ('dv', slice(2 * inner, 3 * inner))):
        report(f's3_mfma_rel.{nm}', got[:, sl], gq[:, sl], 2 ** -6)
    report('s3_mfma_rel.dwth', dwth, wth.grad, 2 ** -6)
    report('s3_mfma_rel.drel', drel[1:].t(), rel.grad, 2 ** -6)


X_CASES = [(3, 20, 7, 2, 32), (2, 100, 33, 8, 64), (1, 64, 256, 8, 64), (2, 33, 31, 4, 32), (2, 70, 64, 3, 64)]


@pytest.mark.parametrize('case', range(len(X_CASES)))
@pytest.mark.parametrize('x3', [False, True])
def test_cross_attention_core(K, O, case, x3):
    B, n, T, heads, dh = X_CASES[case]
    inner = heads * dh
    torch.manual_seed(31 + case)
    rnd = (lambda *s: torch.randn(*s)) if x3 else (lambda *s: bf_round(torch.randn(*s)))
    q = rnd(B, n, heads, dh).requires_grad_(True)
    kv = rnd(B, T, 2, heads, dh).requires_grad_(True)
    nk, nv = rnd(heads, dh).requires_grad_(True), rnd(heads, dh).requires_grad_(True)
    wth = (torch.randn(heads, heads) * 0.5 + torch.eye(heads)).requires_grad_(True)
    mask = torch.rand(B, T) > 0.3
    mask[0] = False                     # a fully masked sample attends only the null key
    if B > 1:
        mask[1, T // 2:] = False
    o_ref = O.attention_core(q, kv[:, :, 0], kv[:, :, 1], nk, nv, wth, mask, dh ** -0.5)
    do = rnd(B, n, heads, dh)
    o_ref.backward(do)
    g = K.x_geom(B, n, T, heads, dh)
    qp = to_bf_pair(q.detach().reshape(B * n, inner).to(DEV), x3)
    kvp = to_bf_pair(kv.detach().reshape(B * T, 2 * inner).to(DEV), x3)
    pk = K.xattn_pack(g, kvp, nk.detach().to(DEV), nv.detach().to(DEV), mask.to(torch.uint8).to(DEV))
    o, P, Pm = K.xattn_fwd(g, qp, pk, wth.detach().to(DEV))
    tag = f'[{case},x3={x3}]'
    val = (lambda p: bf_value(p)) if x3 else (lambda p: p.hi.float())
    report('xattn_fwd' + tag, val(o).reshape(B, n, heads, dh), o_ref.detach(), 5e-5 if x3 else 2 ** -7)
    dop = to_bf_pair(do.reshape(B * n, inner).to(DEV), x3)
    dq, dS, dwth = K.xattn_bwd(g, dop, pk, wth.detach().to(DEV), P)
    report('xattn_dq' + tag, val(dq).reshape(B, n, heads, dh), q.grad, 1e-4 if x3 else 2 ** -6)
    report('xattn_dwth' + tag, dwth, wth.grad, 2e-4 if x3 else 2 ** -6)
    dKp, dVp = K.xattn_kv_grads(g, dS, Pm, qp, dop)
    dkv, dnk, dnv = K.xattn_unpack(g, dKp, dVp, lo=x3)
    report('xattn_dkv' + tag, val(dkv).reshape(B, T, 2, heads, dh), kv.grad, 1e-4 if x3 else 2 ** -6)
    report('xattn_dnull_k' + tag, dnk, nk.grad, 1e-4 if x3 else 2 ** -6)
    report('xattn_dnull_v' + tag, dnv, nv.grad, 1e-4 if x3 else 2 ** -6)
    if not x3 and K.xattn2_supported(g, qp):
        # second design (one wave = 16 queries x all heads, statistics-only forward, recomputing backward)
        o2, stats = K.xattn2_fwd(g, qp, pk, wth.detach().to(DEV))
        report('xattn2_fwd' + tag, o2.hi.float().reshape(B, n, heads, dh), o_ref.detach(), 2 ** -7)
        dq2, dS2, Pm2, dwth2 = K.xattn2_bwd(g, qp, dop, pk, wth.detach().to(DEV), stats)
        report('xattn2_dq' + tag, dq2.hi.float().reshape(B, n, heads, dh), q.grad, 2 ** -6)
        report('xattn2_dwth' + tag, dwth2, wth.grad, 2 ** -6)
        # (xattn2_bwd writes the columns of dS / Pm in its chunk-permuted key order: undone here, and by xattn_unpack below)
        # (... and only the columns up to the last lane group that holds a key exist: compare the keys 0 .. T)
        pos = K.xattn_key_positions(g.JP, DEV)[:T + 1]
        report('xattn2_Pm' + tag, K.xattn_rows(g, Pm2.hi).float().index_select(-1, pos), Pm.hi.float()[..., :T + 1], 2 ** -6)
        report('xattn2_dS' + tag, K.xattn_rows(g, dS2.hi).float().index_select(-1, pos), dS.hi.float()[..., :T + 1], 2 ** -5)
        if dS2.hi.dim() == 5:
            # the chunk-major arrays (what the step uses where the whole-M TN kernel takes them) hold the bits of the row-major ones
            _, dS3, Pm3, _ = K.xattn2_bwd(g, qp, dop, pk, wth.detach().to(DEV), stats, chunk_major=False)
            assert torch.equal(K.xattn_rows(g, dS2.hi), dS3.hi) and torch.equal(K.xattn_rows(g, Pm2.hi), Pm3.hi)
            dKp3, dVp3 = K.xattn_kv_grads(g, dS3, Pm3, qp, dop)
        dKp2, dVp2 = K.xattn_kv_grads(g, dS2, Pm2, qp, dop)
        if dS2.hi.dim() == 5:
            mx = K.xattn_permuted_extent(g)
            assert torch.equal(dKp2[:, :, :mx], dKp3[:, :, :mx]) and torch.equal(dVp2[:, :, :mx], dVp3[:, :, :mx])
        dkv2, dnk2, dnv2 = K.xattn_unpack(g, dKp2, dVp2, lo=False, permuted=True)
        report('xattn2_dkv' + tag, dkv2.hi.float().reshape(B, T, 2, heads, dh), kv.grad, 2 ** -6)
        report('xattn2_dnull_k' + tag, dnk2, nk.grad, 2 ** -6)
        report('xattn2_dnull_v' + tag, dnv2, nv.grad, 2 ** -6)


@pytest.mark.parametrize('M,N,Kd', [(2560, 512, 1536), (4096 + 77, 1536 + 24, 512), (300, 264, 64), (70000, 512, 2752), (2560 * 8, 2752, 1408),
                                    (2560 * 8, 512, 1376), (1000, 600, 96), (777, 512, 160)])       # (the last three: K % 64 == 32, the zeroed half iteration)
def test_gemm_nt_long_k_kernel_equals_the_ring(K, M, N, Kd):
    """gemm_nt_w4k_kernel (four waves of 128x128, K-step 64, two stages with a 1.5-iteration prefetch; forced for every K % 64 == 0
    through tuning key 22 = 2) against the 8-wave ring (key 22 = 1): same accumulation order -> bit-identical, on every epilogue it
    shares with the ring -- fp32 out + bias, bf16 out, bf16 out + GEGLU gate, the GEGLU backward, fp16 operands with an fp16 second
    copy and with the gate -- ragged edges included; and against fp32 torch on the same operands"""
    from nuwa_pytorch_amd import _lib
    L = _lib.lib()
    torch.manual_seed(M % 97)
    a = (torch.randn(M, Kd, device=DEV) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, Kd, device=DEV) * 0.2).to(torch.bfloat16)
    A, W = K.BF(a, None), K.BF(w, None)
    bias = torch.randn(N, device=DEV)
    gate_ok = N % 32 == 0
    u = K.BF((torch.randn(M, 2 * N, device=DEV)).to(torch.bfloat16), None) if gate_ok else None

    def run():
        outs = [K.gemm_nt(A, W, bias=bias, alpha=0.5), K.gemm_nt(A, W, out_bf16=True).hi]
        if gate_ok:
            gg = K.empty_bf((M, N // 2), DEV, lo=False)
            outs += [K.gemm_nt(A, W, out_bf16=True, geglu_out=gg).hi, gg.hi, K.gemm_nt_geglu_bwd(A, W, u, N).hi]
        if K.gemm_nt_f16ops_ok(M, N, Kd, out_bf16=True):
            a16, w16 = a.half(), w.half()
            c = K.gemm_nt_f16ops(a16, w16, out_bf16=True, copy_f16=True)
            outs += [K.gemm_nt_f16ops(a16, w16), c.hi, c.f16]
            if gate_ok:
                outs += list(K.gemm_nt_f16ops(a16, w16, out_bf16=True, gate=True))
        return [o.float().clone() for o in outs]

    try:
        L.amdnuwa_set_tuning(0, 7)                  # the 256x256 tile family for every size (auto keeps small problems on smaller tiles)
        L.amdnuwa_set_tuning(22, 1)
        ref = run()
        L.amdnuwa_set_tuning(22, 2)
        got = run()
    finally:
        L.amdnuwa_set_tuning(22, 0)
        L.amdnuwa_set_tuning(0, 0)
    assert len(ref) == len(got) and all(torch.equal(x, y) for x, y in zip(ref, got)), [i for i, (x, y) in enumerate(zip(ref, got)) if not torch.equal(x, y)]
    report(f'gemm_nt_w4k[{M}x{N}x{Kd}]', got[0], 0.5 * (a.float() @ w.float().t()) + bias, 1e-5)


@pytest.mark.parametrize('R,N1,N2', [(2560 * 4, 1536, 512), (64 * 37, 2752, 512), (2560 * 2, 520, 264), (128, 256, 256), (2560, 512, 1365)])
def test_gemm_tn_four_wave_kernel(K, R, N1, N2):
    """gemm_tn_w4k_kernel (weight gradients: four waves, 64 token rows per iteration; default when there is no token shift and the row
    count is a multiple of 64) against fp64 torch on the same bf16 operands and against the 8-wave ring (tuning key 23 = 1; the split
    boundaries differ, so the comparison is to 2e-6, not bit for bit); bit-repeatable; beta / alpha honoured"""
    from nuwa_pytorch_amd import _lib
    L = _lib.lib()
    torch.manual_seed(R % 89)
    a = K.BF((torch.randn(R, N1, device=DEV) * 0.5).to(torch.bfloat16), None)
    ldb = (N2 + 15) // 16 * 16                           # (an output width that is no multiple of 8: the operand rows are padded, as FF2's gate output)
    bfull = (torch.randn(R, ldb, device=DEV) * 0.5).to(torch.bfloat16)
    b = K.BF(bfull[:, :N2] if ldb != N2 else bfull, None)
    ref = (a.hi.double().t() @ b.hi.double()).float()
    out = torch.empty(N1, N2, device=DEV)
    K.gemm_tn(a, b, out, N2=N2)
    again = torch.empty_like(out)
    K.gemm_tn(a, b, again, N2=N2)
    assert torch.equal(out, again)
    report(f'gemm_tn_w4k[{R},{N1},{N2}]', out, ref, 2e-6)
    try:
        L.amdnuwa_set_tuning(23, 1)
        ring = torch.empty_like(out)
        K.gemm_tn(a, b, ring, N2=N2)
    finally:
        L.amdnuwa_set_tuning(23, 0)
    report(f'gemm_tn_w4k_vs_ring[{R},{N1},{N2}]', out, ring, 2e-6)
    acc = torch.ones(N1, N2, device=DEV)
    K.gemm_tn(a, b, acc, beta=1.0, N2=N2)
    report(f'gemm_tn_w4k_beta[{R},{N1},{N2}]', acc, ref + 1.0, 2e-6)


@pytest.mark.parametrize('B,n,T', [(2, 96, 33), (1, 64, 256), (3, 160, 100), (2, 32, 287)])
def test_cross_attention_bwd_recomputing_key_side(K, O, B, n, T):
    """amdnuwa_xattn2_bwd_rc (query side + recomputing key side, no dS / Pm arrays) against the oracle's autograd and against the
    dS / Pm + batched TN GEMM form it replaces"""
    import ctypes as C
    from nuwa_pytorch_amd import _lib
    heads, dh = 8, 64
    inner = heads * dh
    torch.manual_seed(77 + n)
    rnd = lambda *s: bf_round(torch.randn(*s))
    q = rnd(B, n, heads, dh).requires_grad_(True)
    kv = rnd(B, T, 2, heads, dh).requires_grad_(True)
    nk, nv = rnd(heads, dh).requires_grad_(True), rnd(heads, dh).requires_grad_(True)
    wth = (torch.randn(heads, heads) * 0.5 + torch.eye(heads)).requires_grad_(True)
    mask = torch.rand(B, T) > 0.3
    mask[0] = False
    if B > 1:
        mask[1, T // 2:] = False
    o_ref = O.attention_core(q, kv[:, :, 0], kv[:, :, 1], nk, nv, wth, mask, dh ** -0.5)
    do = rnd(B, n, heads, dh)
    o_ref.backward(do)
    g = K.x_geom(B, n, T, heads, dh)
    assert _lib.lib().amdnuwa_xattn2_bwd_rc_supported(C.byref(g))
    qp = to_bf_pair(q.detach().reshape(B * n, inner).to(DEV), False)
    kvp = to_bf_pair(kv.detach().reshape(B * T, 2 * inner).to(DEV), False)
    dop = to_bf_pair(do.reshape(B * n, inner).to(DEV), False)
    w = wth.detach().to(DEV)
    pk = K.xattn_pack(g, kvp, nk.detach().to(DEV), nv.detach().to(DEV), mask.to(torch.uint8).to(DEV))
    _, stats = K.xattn2_fwd(g, qp, pk, w)
    dq0, dS, Pm, dwth0 = K.xattn2_bwd(g, qp, dop, pk, w, stats)
    dKp0, dVp0 = K.xattn_kv_grads(g, dS, Pm, qp, dop)
    dq, dKp, dVp, dwth = K.xattn2_bwd_rc(g, qp, dop, pk, w, stats)
    tag = f'[{B},{n},{T}]'
    assert torch.equal(dq.hi, dq0.hi) and torch.equal(dwth, dwth0), 'the query side is the same kernel'
    pos = K.xattn_key_positions(g.JP, DEV)[:T + 1]                # (the TN form's rows follow xattn2_bwd's chunk-permuted key order; rows of padding keys are not written)
    report('xattn_rc_dKp_vs_tn' + tag, dKp[:, :, :T + 1], dKp0.index_select(2, pos), 2e-3)           # same bf16 operands, another summation order
    report('xattn_rc_dVp_vs_tn' + tag, dVp[:, :, :T + 1], dVp0.index_select(2, pos), 2e-3)
    dkv, dnk, dnv = K.xattn_unpack(g, dKp, dVp, lo=False)
    report('xattn_rc_dkv' + tag, dkv.hi.float().reshape(B, T, 2, heads, dh), kv.grad, 2 ** -6)
    report('xattn_rc_dnull_k' + tag, dnk, nk.grad, 2 ** -6)
    report('xattn_rc_dnull_v' + tag, dnv, nv.grad, 2 ** -6)
    # repeatable bit for bit (fixed summation order, no atomics)
    _, dKp2, dVp2, _ = K.xattn2_bwd_rc(g, qp, dop, pk, w, stats)
    assert torch.equal(dKp, dKp2) and torch.equal(dVp, dVp2)


@pytest.mark.parametrize('Bq,heads,n,T,dh', [(2, 8, 2560, 256, 64), (3, 4, 100, 200, 32), (1, 2, 71, 300, 64), (5, 8, 640, 129, 64),
                                             (2, 8, 320, 360, 64), (2, 4, 96, 230, 32)])       # (the last two: six / four row fragments per wave of the lean form)
def test_batched_tn_whole_m_kernel_equals_the_tiled_one(K, Bq, heads, n, T, dh):
    """gemm_tn_wm_kernel (one workgroup per (sample, head) owns all JP rows of dK / dV) sums every element in the order of the 128-row
    tiles of gemm_tn_glds_kernel (tuning key 25 = 1): bit-identical, ragged token counts and column counts included"""
    from nuwa_pytorch_amd import _lib
    L = _lib.lib()
    torch.manual_seed(5 + n)
    g = K.x_geom(Bq, n, T, heads, dh)
    inner = heads * dh
    mx = (T + 1 + 7) // 8 * 8
    dS = K.BF(bf_round(torch.randn(Bq, heads, n, g.JP)).to(torch.bfloat16).to(DEV)[..., :mx], None)
    Pm = K.BF(bf_round(torch.rand(Bq, heads, n, g.JP)).to(torch.bfloat16).to(DEV)[..., :mx], None)
    qp = to_bf_pair(torch.randn(Bq * n, inner).to(DEV), False)
    dop = to_bf_pair(torch.randn(Bq * n, inner).to(DEV), False)
    got = K.xattn_kv_grads(g, dS, Pm, qp, dop)
    L.amdnuwa_set_tuning(25, 1)
    try:
        ref = K.xattn_kv_grads(g, dS, Pm, qp, dop)
    finally:
        L.amdnuwa_set_tuning(25, 0)
    for a, b in zip(got, ref):
        assert torch.equal(a[:, :, :mx], b[:, :, :mx])
    # round 6: whole 32-row K-steps take the lean form (gemm_tn_wmf_kernel); tuning key 25 = 3 keeps the general whole-M kernel
    L.amdnuwa_set_tuning(25, 3)
    try:
        gen = K.xattn_kv_grads(g, dS, Pm, qp, dop)
    finally:
        L.amdnuwa_set_tuning(25, 0)
    for a, b in zip(got, gen):
        assert torch.equal(a[:, :, :mx], b[:, :, :mx])
    # the same operands in planes of 32 columns (a_chunk32: how xattn2_bwd writes them for this kernel)
    assert K.xattn_chunk_major_ok(g)
    cm = lambda t: K.BF(torch.nn.functional.pad(t.hi, (0, g.JP - mx)).reshape(Bq, heads, n, g.JP // 32, 32).permute(0, 1, 3, 2, 4).contiguous(), None)
    for a, b in zip(K.xattn_kv_grads(g, cm(dS), cm(Pm), qp, dop), got):
        assert torch.equal(a[:, :, :mx], b[:, :, :mx])
    want = torch.einsum('bhnj,bnhd->bhjd', dS.hi.float(), qp.hi.float().reshape(Bq, n, heads, dh)) * g.scale
    report(f'tn_whole_m[{Bq},{heads},{n},{T}]', got[0][:, :, :mx], want, 2e-3)


@pytest.mark.parametrize('shape,kern,dil,n', [((2, 16, 16), (5, 3, 3), (1, 1, 1), None), ((3, 16, 16), (3, 3, 3), (4, 4, 4), 300),
                                              ((5, 16, 16), (5, 3, 3), (2, 2, 2), 1 + 4 * 256 + 100)])
def test_sparse3dna_bwd_recomputing_key_side(K, O, shape, kern, dil, n):
    """the recomputing key-side backward (tuning key 4 = 4; no ds / P' workspace: statistics + delta from the query side, scores and the
    head mix redone per plane) against the default workspace form and against the oracle's autograd"""
    from nuwa_pytorch_amd import _lib
    L = _lib.lib()
    heads, dh = 8, 64
    N = shape[0] * shape[1] * shape[2]
    n = N if n is None else n
    B, inner = 2, heads * dh
    torch.manual_seed(21)
    qkv = bf_round(torch.randn(B, n, 3, heads, dh)).requires_grad_(True)
    wth = (torch.randn(heads, heads) * 0.5 + torch.eye(heads)).requires_grad_(True)
    idx = O.neighbor_table(shape, kern, dil, causal=True)
    o_ref = O.sparse3dna_core(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], wth, idx, dh ** -0.5)
    do = bf_round(torch.randn_like(o_ref))
    o_ref.backward(do)
    g = K.s3_geom(B, n, shape, kern, dil, heads, dh)
    qkvp = to_bf_pair(qkv.detach().reshape(B * n, 3 * inner).to(DEV), False)
    dop = to_bf_pair(do.reshape(B * n, inner).to(DEV), False)
    old, dwth_o, _ = K.sparse3dna_bwd(g, qkvp, wth.detach().to(DEV), dop)
    try:
        L.amdnuwa_set_tuning(4, 4)
        new, dwth_n, _ = K.sparse3dna_bwd(g, qkvp, wth.detach().to(DEV), dop)
    finally:
        L.amdnuwa_set_tuning(4, 0)
    gq = qkv.grad.reshape(B * n, 3 * inner)
    tag = f'[{shape},{dil},{n}]'
    assert torch.equal(new.hi[:, :inner], old.hi[:, :inner])                   # dq: the query side is the same arithmetic
    report('s3_bwd_rc.dwth_vs_workspace' + tag, dwth_n, dwth_o, 1e-6)
    for nm, sl in (('dk', slice(inner, 2 * inner)), ('dv', slice(2 * inner, 3 * inner))):
        report(f's3_bwd_rc.{nm}_vs_workspace' + tag, new.hi[:, sl].float(), old.hi[:, sl].float(), 2 ** -6)
        report(f's3_bwd_rc.{nm}_vs_oracle' + tag, new.hi[:, sl].float(), gq[:, sl], 2 ** -6)


# ---------------------------------------------------------------------------------------------------
# the fp16-operand forward cores of the 'bf16x3-fwd' mode
# ---------------------------------------------------------------------------------------------------

def _f16_pair(t):
    """fp32 CPU tensor -> BF(hi = bf16 copy, None, f16 = fp16 copy) on the device + the fp16-rounded fp32 values"""
    h = t.half()
    return K_BF(t.to(torch.bfloat16).to(DEV), None, h.to(DEV)), h.float()


def K_BF(hi, lo, f16):
    from nuwa_pytorch_amd.kernels import BF
    return BF(hi.contiguous(), lo, f16.contiguous())


@pytest.mark.parametrize('shape,kern,dil,n', [((2, 16, 16), (5, 3, 3), (1, 1, 1), None), ((3, 16, 16), (3, 3, 3), (4, 4, 4), 300),
                                              ((4, 16, 16), (5, 3, 3), (2, 2, 2), 1 + 3 * 256 + 37)])
def test_sparse3dna_fwd_f16_core(K, O, shape, kern, dil, n):
    """the MFMA band kernel on fp16 operands (single fp16 MFMAs, hi + lo output) against the oracle on the SAME fp16-rounded q, k, v:
    what is left is the fp16 rounding of the probabilities before P'V (2^-11 per term) and fp32 accumulation order"""
    heads, dh = 8, 64
    N = shape[0] * shape[1] * shape[2]
    n = N if n is None else n
    B, inner = 2, heads * dh
    torch.manual_seed(3)
    qkv = torch.randn(B * n, 3 * inner)
    qkvp, qr = _f16_pair(qkv)
    wth = torch.randn(heads, heads) * 0.5 + torch.eye(heads)
    idx = O.neighbor_table(shape, kern, dil, causal=True)
    q3 = qr.reshape(B, n, 3, heads, dh)
    o_ref = O.sparse3dna_core(q3[:, :, 0], q3[:, :, 1], q3[:, :, 2], wth, idx, dh ** -0.5)
    g = K.s3_geom(B, n, shape, kern, dil, heads, dh)
    assert K.s3_f16_supported(g)
    o = K.sparse3dna_fwd(g, qkvp, wth.to(DEV))
    assert o.lo is not None
    report(f's3_fwd_f16[{shape},{dil},{n}]', bf_value(o).reshape(B, n, heads, dh), o_ref, 6e-4)
    # and against the 3-MFMA (hi + lo) kernel on the full-precision inputs: the difference IS the fp16 rounding of q, k, v, P
    o3 = K.sparse3dna_fwd(g, to_bf_pair(qkv.to(DEV), True), wth.to(DEV))
    report(f's3_fwd_f16_vs_x3[{shape},{dil},{n}]', bf_value(o), bf_value(o3), 3e-3)


@pytest.mark.parametrize('rows', [2, 4])
@pytest.mark.parametrize('shape,kern,dil,n,bias', [((2, 16, 16), (5, 3, 3), (1, 1, 1), None, False), ((3, 16, 16), (3, 3, 3), (4, 4, 4), 300, False),
                                                   ((5, 16, 16), (5, 3, 3), (2, 2, 2), 1 + 4 * 256 + 100, False), ((3, 16, 16), (5, 3, 3), (2, 1, 4), 530, True),
                                                   ((2, 16, 16), (3, 2, 3), (1, 2, 1), 1 + 256 + 16, True), ((10, 16, 16), (5, 3, 3), (4, 4, 4), None, False)])
def test_sparse3dna_fwd_multi_row_tiles(K, O, rows, shape, kern, dil, n, bias):
    """the multi-row tiles of the MFMA forward (tuning key 16: ROWS query rows of one residue class of y per workgroup, every key /
    value row staged once per tile) against the oracle and against the one-row kernel, bf16 and fp16 operand forms, with partial
    last rows / frames, per-axis dilations and the relative-position bias"""
    from nuwa_pytorch_amd import _lib
    L = _lib.lib()
    heads, dh = 8, 64
    N = shape[0] * shape[1] * shape[2]
    n = N if n is None else n
    B, inner = 2, heads * dh
    J = kern[0] * kern[1] * kern[2] + 1
    torch.manual_seed(31)
    qkv = torch.randn(B * n, 3 * inner)
    wth = torch.randn(heads, heads) * 0.5 + torch.eye(heads)
    rel = torch.randn(heads, J - 1) * 0.7 if bias else None
    rel_dev = torch.cat((torch.zeros(1, heads), rel.t()), 0).contiguous().to(DEV) if bias else None
    idx = O.neighbor_table(shape, kern, dil, causal=True)
    g = K.s3_geom(B, n, shape, kern, dil, heads, dh)
    tag = f'[{rows},{shape},{dil},{n}]'
    for f16 in (False, True):
        if f16:
            qkvp, qr = _f16_pair(qkv)
        else:
            qr = bf_round(qkv)
            qkvp = to_bf_pair(qr.to(DEV), False)
        q3 = qr.reshape(B, n, 3, heads, dh)
        o_ref = O.sparse3dna_core(q3[:, :, 0], q3[:, :, 1], q3[:, :, 2], wth, idx, dh ** -0.5, rel_pos_bias=rel)
        try:
            L.amdnuwa_set_tuning(16, 1)
            o1 = K.sparse3dna_fwd(g, qkvp, wth.to(DEV), rel_bias=rel_dev)
            L.amdnuwa_set_tuning(16, rows)
            o2 = K.sparse3dna_fwd(g, qkvp, wth.to(DEV), rel_bias=rel_dev)
            o2b = K.sparse3dna_fwd(g, qkvp, wth.to(DEV), rel_bias=rel_dev)
        finally:
            L.amdnuwa_set_tuning(16, 0)
        assert torch.equal(o2.hi, o2b.hi)
        v1, v2 = (bf_value(o1), bf_value(o2)) if f16 else (o1.hi.float(), o2.hi.float())
        report(f's3_tile.vs_oracle[f16={f16}]' + tag, v2.reshape(B, n, heads, dh), o_ref, 6e-4 if f16 else 2 ** -7)
        # same scores bit for bit; the apply pass pairs its key rows differently inside an MFMA: fp32 summation order, i.e. at most
        # one bf16 step on the hi part (2^-8 of the value) / a few 2^-17 on the hi + lo pair
        report(f's3_tile.vs_one_row[f16={f16}]' + tag, v2, v1, 2e-5 if f16 else 2 ** -8)


@pytest.mark.parametrize('B,n,T', [(2, 100, 33), (1, 64, 256), (2, 2560, 256)])
def test_cross_attention_fwd_f16_core(K, O, B, n, T):
    """the xattn4 core on fp16 operands (fp16 K / V images from xattn_pack, single fp16 MFMAs incl. the head mix, hi + lo output,
    statistics for the bf16 backward) against the oracle on the same fp16-rounded inputs, and its statistics against the bf16 kernel's"""
    heads, dh = 8, 64
    inner = heads * dh
    torch.manual_seed(7)
    q, kv = torch.randn(B * n, inner), torch.randn(B * T, 2 * inner)
    nk, nv = torch.randn(heads, dh), torch.randn(heads, dh)
    wth = torch.randn(heads, heads) * 0.5 + torch.eye(heads)
    mask = torch.rand(B, T) > 0.3
    mask[0] = False
    qp, qr = _f16_pair(q)
    kvp, kvr = _f16_pair(kv)
    kv4 = kvr.reshape(B, T, 2, heads, dh)
    o_ref = O.attention_core(qr.reshape(B, n, heads, dh), kv4[:, :, 0], kv4[:, :, 1], nk.half().float(), nv.half().float(), wth, mask, dh ** -0.5)
    g = K.x_geom(B, n, T, heads, dh)
    pk = K.xattn_pack(g, kvp, nk.to(DEV), nv.to(DEV), mask.to(torch.uint8).to(DEV))
    o, stats = K.xattn2_fwd_f16(g, qp, pk, wth.to(DEV))
    report(f'xattn_fwd_f16[{B},{n},{T}]', bf_value(o).reshape(B, n, heads, dh), o_ref, 1e-3)
    # the bf16 kernel on the hi images: same statistics up to the operand rounding (row max in the log2 domain, 1 / row sum)
    o2, stats2 = K.xattn2_fwd(g, K.BF(qp.hi, None), pk, wth.to(DEV))
    report(f'xattn_fwd_f16.stats[{B},{n},{T}]', stats, stats2, 3e-2)
    report(f'xattn_fwd_f16_vs_bf16[{B},{n},{T}]', bf_value(o), o2.hi.float(), 3e-2)


@pytest.mark.parametrize('B,n,T', [(2, 100, 33), (1, 64, 256), (2, 2560, 256), (3, 70, 1), (2, 130, 64), (1, 200, 300)])
@pytest.mark.parametrize('f16', [True, False])
def test_cross_attention_xattn6_fwd(K, O, B, n, T, f16):
    """the third-design forward core (images in LDS order, K pre-scaled into the log2 domain, the key mask as the C operand of the score MFMAs,
    the null key as a rank-one term -- any context length) against the oracle on the same 16-bit-rounded inputs; its statistics
    reproduce the probabilities (checked through the old kernel's statistics: same normaliser up to the operand rounding)"""
    heads, dh = 8, 64
    inner = heads * dh
    torch.manual_seed(11)
    q, kv = torch.randn(B * n, inner), torch.randn(B * T, 2 * inner)
    nk, nv = torch.randn(heads, dh), torch.randn(heads, dh)
    wth = torch.randn(heads, heads) * 0.5 + torch.eye(heads)
    mask = torch.rand(B, T) > 0.3
    mask[0] = False                     # a fully masked sample attends only the null key
    dt = torch.float16 if f16 else torch.bfloat16
    q16, kv16 = q.to(dt).to(DEV), kv.to(dt).to(DEV)
    kv4 = kv16.float().cpu().reshape(B, T, 2, heads, dh)
    o_ref = O.attention_core(q16.float().cpu().reshape(B, n, heads, dh), kv4[:, :, 0], kv4[:, :, 1], nk, nv, wth, mask, dh ** -0.5)
    g = K.x_geom(B, n, T, heads, dh)
    assert K.xattn6_supported(g)
    pk = K.xattn6_pack(g, kv16, mask.to(torch.uint8).to(DEV))
    o, stats = K.xattn6_fwd(g, q16, pk, nk.to(DEV), nv.to(DEV), wth.to(DEV))
    tag = f'[{B},{n},{T},f16={f16}]'
    report('xattn6_fwd' + tag, bf_value(o).reshape(B, n, heads, dh), o_ref, 1e-3 if f16 else 2 ** -7)
    # the fp16 copy of the output (what the two-MFMA to_out product reads)
    o2, _ = K.xattn6_fwd(g, q16, pk, nk.to(DEV), nv.to(DEV), wth.to(DEV), o_f16=True)
    assert torch.equal(o2.hi, o.hi)
    report('xattn6_fwd.o_f16' + tag, o2.f16.float().reshape(B, n, heads, dh), o_ref, 1e-3 if f16 else 2 ** -7)
    # statistics: log2-domain normaliser  log2(sum_j exp2(s_j)) = m + log2(1 / il)  against fp32 torch on the same operands
    sc = torch.einsum('bihd,bjhd->bhij', q16.float().cpu().reshape(B, n, heads, dh), kv4[:, :, 0]) * (dh ** -0.5)
    sn = torch.einsum('bihd,hd->bhi', q16.float().cpu().reshape(B, n, heads, dh), nk) * (dh ** -0.5)
    sc = sc.masked_fill(~mask[:, None, None, :], float('-inf'))
    lse = torch.logsumexp(torch.cat([sn[..., None], sc], dim=-1), dim=-1) * math.log2(math.e)
    got = stats[..., 0].cpu() - torch.log2(stats[..., 1].cpu())
    assert (got - lse).abs().max().item() < (2e-2 if f16 else 1e-1), (got - lse).abs().max().item()
    if T + 1 <= 287:
        # and the old pair (xattn_pack + xattn4): same result up to the rounding of K * c1 / the null key's fp32 treatment
        kvp = K.BF(kv16.to(torch.bfloat16) if f16 else kv16, None, kv16 if f16 else None)
        if f16:
            pko = K.xattn_pack(g, kvp, nk.to(DEV), nv.to(DEV), mask.to(torch.uint8).to(DEV))
            o4, _ = K.xattn2_fwd_f16(g, K.BF(q16.to(torch.bfloat16), None, q16), pko, wth.to(DEV))
            report('xattn6_vs_xattn4' + tag, bf_value(o), bf_value(o4), 2e-3)


@pytest.mark.parametrize('B,n,T', [(2, 100, 33), (2, 64, 256), (2, 2560, 256), (3, 70, 1), (2, 130, 64), (4, 300, 200)])
def test_cross_attention_xattn6_bwd(K, O, B, n, T):
    """the query side of the recomputing backward on xattn6 images against the oracle's gradients (through the batched dK / dV products and
    xattn_unpack) and against xattn2_bwd on the same statistics (same rounding points; the score scale is one fma instead of mul + add)"""
    heads, dh = 8, 64
    inner = heads * dh
    torch.manual_seed(13)
    q = bf_round(torch.randn(B, n, heads, dh)).requires_grad_(True)
    kv = bf_round(torch.randn(B, T, 2, heads, dh)).requires_grad_(True)
    nk, nv = bf_round(torch.randn(heads, dh)).requires_grad_(True), bf_round(torch.randn(heads, dh)).requires_grad_(True)
    wth = (torch.randn(heads, heads) * 0.5 + torch.eye(heads)).requires_grad_(True)
    mask = torch.rand(B, T) > 0.3
    mask[0] = False                     # a fully masked sample attends only the null key: its dq is exactly zero
    o_ref = O.attention_core(q, kv[:, :, 0], kv[:, :, 1], nk, nv, wth, mask, dh ** -0.5)
    do = bf_round(torch.randn(B, n, heads, dh))
    o_ref.backward(do)
    g = K.x_geom(B, n, T, heads, dh)
    if not K.xattn6_bwd_ok(g):
        pytest.skip('whole-M chunk-major TN product not taken for this shape')
    qp = K.BF(q.detach().reshape(B * n, inner).to(torch.bfloat16).to(DEV), None)
    kvp = K.BF(kv.detach().reshape(B * T, 2 * inner).to(torch.bfloat16).to(DEV), None)
    dop = K.BF(do.reshape(B * n, inner).to(torch.bfloat16).to(DEV), None)
    m8 = mask.to(torch.uint8).to(DEV)
    w = wth.detach().to(DEV)
    pk6 = K.xattn6_pack(g, kvp.hi, m8)
    o, stats = K.xattn6_fwd(g, qp.hi, pk6, nk.detach().to(DEV), nv.detach().to(DEV), w, lo=False)
    pkb = K.xattn6_pack_bwd(g, kvp.hi, nk.detach().to(DEV), nv.detach().to(DEV), m8)
    dq, dS, Pm, dwth = K.xattn6_bwd(g, qp, dop, pkb, w, stats)
    tag = f'[{B},{n},{T}]'
    report('xattn6_bwd.dq' + tag, dq.hi.float().reshape(B, n, heads, dh), q.grad, 2 ** -6)
    report('xattn6_bwd.dwth' + tag, dwth, wth.grad, 2 ** -6)
    dKp, dVp = K.xattn_kv_grads(g, dS, Pm, qp, dop)
    dkv, dnk, dnv = K.xattn_unpack(g, dKp, dVp, lo=False, permuted=True, null_last=True)
    report('xattn6_bwd.dkv' + tag, dkv.hi.float().reshape(B, T, 2, heads, dh), kv.grad, 2 ** -6)
    report('xattn6_bwd.dnull_k' + tag, dnk, nk.grad, 2 ** -6)
    report('xattn6_bwd.dnull_v' + tag, dnv, nv.grad, 2 ** -6)
    # the second design on the same statistics
    pko = K.xattn_pack(g, kvp, nk.detach().to(DEV), nv.detach().to(DEV), m8)
    dq2, dS2, Pm2, dwth2 = K.xattn2_bwd(g, qp, dop, pko, w, stats, chunk_major=True)
    report('xattn6_bwd.vs_xattn2.dq' + tag, dq.hi.float(), dq2.hi.float(), 2 ** -7)
    dKp2, dVp2 = K.xattn_kv_grads(g, dS2, Pm2, qp, dop)
    dkv2, dnk2, dnv2 = K.xattn_unpack(g, dKp2, dVp2, lo=False, permuted=True)
    report('xattn6_bwd.vs_xattn2.dkv' + tag, dkv.hi.float(), dkv2.hi.float(), 2 ** -7)
    report('xattn6_bwd.vs_xattn2.dnull_k' + tag, dnk, dnk2, 2 ** -7)
    report('xattn6_bwd.vs_xattn2.dnull_v' + tag, dnv, dnv2, 2 ** -7)
    report('xattn6_bwd.vs_xattn2.dwth' + tag, dwth, dwth2, 2 ** -8)


def test_gemm_nt_fp16_operands_and_ln_fp16_copy(K):
    """the FeedForward forward of 'bf16x3-fwd': LayerNorm stores a bf16 + fp16 copy pair, FF1 runs on the fp16 MFMA with the gate in its
    epilogue (u bf16 for the backward, gate output as fp16 + bf16 copies), FF2 on the fp16 MFMA to fp32 -- against fp64 on the same
    fp16 operand values"""
    torch.manual_seed(4)
    R, D, FP = 16384 + 40, 512, 1376
    x = torch.randn(R, D, device=DEV) * 1.5 + 0.2
    w, b = torch.randn(D, device=DEV), torch.randn(D, device=DEV)
    h, m, r, _ = K.ln_fwd(x, w, b, f16=True)
    h2, _, _, _ = K.ln_fwd(x, w, b)
    assert h.lo is None and h.f16.dtype == torch.float16 and torch.equal(h.hi, h2.hi)
    ref_h = torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double())
    report('ln_fwd.f16_copy', h.f16.float(), ref_h.float(), 2 ** -11)
    w1 = (torch.randn(2 * FP, D, device=DEV) * 0.2).half()
    w2 = (torch.randn(D, FP, device=DEV) * 0.2).half()
    assert K.gemm_nt_f16ops_ok(R, 2 * FP, D, out_bf16=True, gate=True) and K.gemm_nt_f16ops_ok(R, D, FP, out_bf16=False)
    u, gg16, ggb = K.gemm_nt_f16ops(h.f16, w1, out_bf16=True, gate=True)
    u_ref = h.f16.double() @ w1.double().t()
    report('gemm_f16ops.u', u.float(), u_ref.float(), 2 ** -8)
    ud = K.geglu_deinterleave(u_ref, FP, dim=1)
    g_ref = (ud[:, :FP] * F.gelu(ud[:, FP:])).float()
    report('gemm_f16ops.gate_f16', gg16.float(), g_ref, 2 ** -10)
    report('gemm_f16ops.gate_bf16', ggb.float(), g_ref, 2 ** -8)
    y = K.gemm_nt_f16ops(gg16, w2)
    report('gemm_f16ops.y_f32', y, (gg16.double() @ w2.double().t()).float(), 2e-6)
    assert not K.gemm_nt_f16ops_ok(20, 96, 64, out_bf16=False)           # the few-row decode shapes stay on the hi + lo kernels


def test_gemm_nt_f16_second_copy(K):
    """projection GEMM of the 'bf16x3-fwd' mode: hi + lo operands, output = bf16 copy + fp16 copy of the SAME fp32 accumulator (on the
    256x256 ring in the epilogue; elsewhere product + conversion pass)"""
    torch.manual_seed(2)
    for M, N, Kd in ((16384, 1536, 512), (300, 96, 64)):
        a, b = torch.randn(M, Kd) * 0.5, torch.randn(N, Kd) * 0.3
        A, Bm = to_bf_pair(a.to(DEV), True), to_bf_pair(b.to(DEV), True)
        ref = (bf_value(A).double() @ bf_value(Bm).double().t()).float()
        out = K.gemm_nt(A, Bm, out_bf16=True, out_f16=True)
        assert out.lo is None and out.f16.dtype == torch.float16
        full = K.gemm_nt(A, Bm, out_bf16=True)
        assert torch.equal(out.hi, full.hi)
        report(f'gemm_f16_copy[{M},{N}]', out.f16.float(), ref, 2 ** -11)
        assert torch.equal(out.f16, bf_value(full).half())


@pytest.mark.parametrize('M,N,Kd', [(16384 + 40, 512, 512), (2560, 8192, 512), (300, 520, 96), (5000, 1024, 1376)])
def test_gemm_nt_two_mfma_form(K, M, N, Kd):
    """fp16 activation x fp16 (hi + lo) weight, two fp16 MFMAs per product on the hi + lo ring (amdnuwa_gemm_desc.ab_f16 with Blo): against
    fp64 on the same operand values (what is left is the fp32 accumulation), against the EXACT fp32 weight (the pair carries it to ~22
    bits), with a bias, and with the bf16 + fp16 output pair"""
    torch.manual_seed(M + N)
    a16 = (torch.randn(M, Kd, device=DEV) * 0.7).half()
    w = torch.randn(N, Kd, device=DEV) * 0.05
    bias = torch.randn(N, device=DEV)
    wp = K.f16_pair(w)
    assert wp[0].dtype == torch.float16 and wp[1].dtype == torch.float16
    report(f'f16_pair[{N},{Kd}]', wp[0].double() + wp[1].double(), w.double(), 2 ** -20)
    assert K.gemm_nt_f16x2_ok(M, N, Kd, out_bf16=False) and K.gemm_nt_f16x2_ok(M, N, Kd, out_bf16=True)
    ref = a16.double() @ (wp[0].double() + wp[1].double()).t()
    y = K.gemm_nt_f16x2(a16, wp, bias=bias)
    assert y.dtype == torch.float32 and tuple(y.shape) == (M, N)
    report(f"gemm_x2.f32[{M},{N},{Kd}]", y, (ref + bias.double()).float(), 3e-6)
    report(f'gemm_x2.f32_vs_exact_w[{M},{N},{Kd}]', y, (a16.double() @ w.double().t() + bias.double()).float(), 4e-6)
    assert torch.equal(y, K.gemm_nt_f16x2(a16, wp, bias=bias))                      # no atomics, fixed order
    o = K.gemm_nt_f16x2(a16, wp, out_bf16=True, copy_f16=True)
    assert o.lo is None and o.hi.dtype == torch.bfloat16 and o.f16.dtype == torch.float16
    report(f'gemm_x2.bf16[{M},{N},{Kd}]', o.hi.float(), ref.float(), 2 ** -8)
    report(f'gemm_x2.f16_copy[{M},{N},{Kd}]', o.f16.float(), ref.float(), 2 ** -11)
    # the three-MFMA product of the same values (a16 as an exact hi + lo pair, the weight as a bf16 pair): the two forms agree to the
    # weight pair's precision
    A3 = K.BF(a16.to(torch.bfloat16), (a16.float() - a16.to(torch.bfloat16).float()).to(torch.bfloat16))
    y3 = K.gemm_nt(A3, to_bf_pair(w, True), bias=bias)
    report(f'gemm_x2_vs_x3[{M},{N},{Kd}]', y, y3, 3e-5)
    assert not K.gemm_nt_f16x2_ok(20, 512, 512, out_bf16=False)                     # the few-row decode shapes stay on the hi + lo kernels


def test_fp16_cores_hand_over_an_fp16_copy(K):
    """o_f16: both fp16 forward cores write o as a bf16 copy (what the backward reads: unchanged, bit for bit) + the fp16 rendering of the
    SAME fp32 value in place of the bf16 residual -- the A operand of the two-MFMA to_out product"""
    heads, dh, B = 8, 64, 2
    inner = heads * dh
    torch.manual_seed(5)
    wth = (torch.randn(heads, heads) * 0.5 + torch.eye(heads)).to(DEV)
    shape, kern, dil, n = (3, 16, 16), (5, 3, 3), (2, 2, 2), 1 + 2 * 256 + 77
    qkvp, _ = _f16_pair(torch.randn(B * n, 3 * inner))
    g = K.s3_geom(B, n, shape, kern, dil, heads, dh)
    o = K.sparse3dna_fwd(g, qkvp, wth)
    o2 = K.sparse3dna_fwd(g, qkvp, wth, o_f16=True)
    assert o2.lo is None and o2.f16.dtype == torch.float16 and torch.equal(o.hi, o2.hi)
    report('s3_fwd_f16.o_f16', o2.f16.float(), bf_value(o), 2 ** -11)
    n, T = 700, 256
    qp, _ = _f16_pair(torch.randn(B * n, inner))
    kvp, _ = _f16_pair(torch.randn(B * T, 2 * inner))
    gx = K.x_geom(B, n, T, heads, dh)
    mask = (torch.rand(B, T) > 0.3).to(torch.uint8).to(DEV)
    pk = K.xattn_pack(gx, kvp, torch.randn(heads, dh, device=DEV), torch.randn(heads, dh, device=DEV), mask)
    x, st = K.xattn2_fwd_f16(gx, qp, pk, wth)
    x2, st2 = K.xattn2_fwd_f16(gx, qp, pk, wth, o_f16=True)
    assert x2.lo is None and torch.equal(x.hi, x2.hi) and torch.equal(st, st2)
    report('xattn_fwd_f16.o_f16', x2.f16.float(), bf_value(x), 2 ** -11)


# ---------------------------------------------------------------------------------------------------
# full-size (BASELINE cfg 3 geometry) size-independent properties
# ---------------------------------------------------------------------------------------------------

def test_sparse3dna_fullsize_causality_and_determinism(K):
    """cfg 3 geometry (10x16x16 tokens, kernel (5,3,3), dilation 2, 8 heads x 64): (i) bit-reproducible
    fwd/bwd across two runs (no atomics on the path), (ii) causal: changing token p leaves every
    output row <= p untouched, bit for bit."""
    shape, kern, dil, heads, dh = (10, 16, 16), (5, 3, 3), (2, 2, 2), 8, 64
    B, n = 1, 2560
    inner = heads * dh
    torch.manual_seed(77)
    qkv = torch.randn(B * n, 3 * inner, device=DEV)
    wth = (torch.randn(heads, heads) * 0.3 + torch.eye(heads)).to(DEV)
    g = K.s3_geom(B, n, shape, kern, dil, heads, dh)
    p1 = to_bf_pair(qkv, False)
    o1 = K.sparse3dna_fwd(g, p1, wth).hi.clone()
    o2 = K.sparse3dna_fwd(g, p1, wth).hi
    assert torch.equal(o1, o2)
    do = to_bf_pair(torch.randn(B * n, inner, device=DEV), False)
    d1, w1, _ = K.sparse3dna_bwd(g, p1, wth, do)
    d2, w2, _ = K.sparse3dna_bwd(g, p1, wth, do)
    assert torch.equal(d1.hi, d2.hi) and torch.equal(w1, w2)
    cut = 1500
    qkv2 = qkv.clone()
    qkv2[cut:] += 1.0
    o3 = K.sparse3dna_fwd(g, to_bf_pair(qkv2, False), wth).hi
    assert torch.equal(o1[:cut], o3[:cut])
    assert not torch.equal(o1[cut:], o3[cut:])
    assert bool(torch.isfinite(o1.float()).all()) and bool(torch.isfinite(d1.hi.float()).all())


def test_attention_kernels_bit_reproducible_with_coresident_workgroups(K):
    """the attention kernels at a batch that puts two workgroups on every CU and runs several rounds of them: 6 runs on the same inputs,
    every output bit for bit (the one-sample determinism test above never has two workgroups on a CU).  Round 4: the head mix of the two-row
    forward tile failed exactly this -- a few wrong P' entries per launch, different ones every run (csrc/sparse3dna.hip, lds_store8_done)"""
    from nuwa_pytorch_amd import _lib
    L = _lib.lib()
    heads, dh, B, n = 8, 64, 16, 2560
    inner = heads * dh
    torch.manual_seed(0)
    wth = (torch.randn(heads, heads) * 0.5 + torch.eye(heads)).to(DEV)

    def outs(o):
        res = []
        for x in (o if isinstance(o, (tuple, list)) and not isinstance(o, K.BF) else (o,)):
            res += [t for t in ((x.hi, x.lo, x.f16) if isinstance(x, K.BF) else (x,)) if torch.is_tensor(t)]
        return res

    def stable(name, fn):
        ref = [t.clone() for t in outs(fn())]
        for _ in range(5):
            for a, b in zip(outs(fn()), ref):
                assert torch.equal(a, b), f'{name}: not bit-reproducible'

    for dil in ((1, 1, 1), (2, 2, 2), (4, 4, 4)):
        qkv = torch.randn(B * n, 3 * inner, device=DEV)
        g = K.s3_geom(B, n, (10, 16, 16), (5, 3, 3), dil, heads, dh)
        pbf, p16 = K.BF(qkv.to(torch.bfloat16), None), K.BF(qkv.to(torch.bfloat16), None, qkv.half())
        dO = K.BF(torch.randn(B * n, inner, device=DEV).to(torch.bfloat16), None)
        for rows in (1, 2):
            try:
                L.amdnuwa_set_tuning(16, rows)
                stable(f'3DNA fwd bf16 dil {dil[0]} rows {rows}', lambda: K.sparse3dna_fwd(g, pbf, wth))
                stable(f'3DNA fwd fp16 dil {dil[0]} rows {rows}', lambda: K.sparse3dna_fwd(g, p16, wth))
            finally:
                L.amdnuwa_set_tuning(16, 0)
        stable(f'3DNA bwd dil {dil[0]}', lambda: K.sparse3dna_bwd(g, pbf, wth, dO))
    T = 256
    q, kv = torch.randn(B * n, inner, device=DEV), torch.randn(B * T, 2 * inner, device=DEV)
    gx = K.x_geom(B, n, T, heads, dh)
    mask = (torch.rand(B, T, device=DEV) > 0.2).to(torch.uint8)
    nk, nv = torch.randn(heads, dh, device=DEV), torch.randn(heads, dh, device=DEV)
    q16, kv16 = K.BF(q.to(torch.bfloat16), None, q.half()), K.BF(kv.to(torch.bfloat16), None, kv.half())
    pk16 = K.xattn_pack(gx, kv16, nk, nv, mask)
    stable('cross attention fwd fp16', lambda: K.xattn2_fwd_f16(gx, q16, pk16, wth))
    qb = K.BF(q.to(torch.bfloat16), None)
    pkb = K.xattn_pack(gx, K.BF(kv.to(torch.bfloat16), None), nk, nv, mask)
    stable('cross attention fwd bf16', lambda: K.xattn2_fwd(gx, qb, pkb, wth))
    _, stats = K.xattn2_fwd(gx, qb, pkb, wth)
    dO = K.BF(torch.randn(B * n, inner, device=DEV).to(torch.bfloat16), None)
    # (chunk-major dS / Pm: the lane groups of the last chunk that hold no key write nothing -- compare the columns that exist)
    rows = lambda r: (r[0], K.xattn_rows(gx, r[1].hi), K.xattn_rows(gx, r[2].hi), r[3])
    stable('cross attention bwd', lambda: rows(K.xattn2_bwd(gx, qb, dO, pkb, wth, stats)))
    stable('cross attention bwd, row-major dS / Pm', lambda: K.xattn2_bwd(gx, qb, dO, pkb, wth, stats, chunk_major=False))


@pytest.mark.parametrize('R,C,Kd', [(1000, 8192, 512), (2560, 512, 256), (300, 192, 64)])
def test_fused_linear_cross_entropy(R, C, Kd):
    """to_logits + cross entropy without the fp32 logits (np.py:1958-1963): loss, dlogits against torch on the same bf16 operands,
    against the unfused libamdnuwa pair (gemm_nt + ce_fwd), a ragged last row tile, and the NaN flag for an id outside the vocabulary"""
    from nuwa_pytorch_amd import kernels as K
    g = torch.Generator().manual_seed(R + C)
    h = (torch.randn(R, Kd, generator=g) * 0.8).to(torch.bfloat16)
    w = (torch.randn(C, Kd, generator=g) * (3.0 / Kd ** 0.5)).to(torch.bfloat16)
    t = torch.randint(0, C, (R,), generator=g)
    t[0], t[1], t[-1] = 0, C - 1, C - 1
    logits = h.float() @ w.float().t()
    ref_loss = torch.nn.functional.cross_entropy(logits, t)
    ref_dl = (logits.softmax(-1) - torch.nn.functional.one_hot(t, C).float()) / R
    hb, wb = K.BF(h.to(DEV), None), K.BF(w.to(DEV), None)
    out = K.linear_ce(hb, wb, t.to(DEV), 1.0 / R)
    assert out is not None
    loss, dl = out
    report(f'linear_ce[{R},{C}].loss', loss.reshape(1), ref_loss.reshape(1), 2e-6)
    report(f'linear_ce[{R},{C}].dlogits', dl.hi.float(), ref_dl, 2 ** -8)
    lg = K.gemm_nt(hb, wb)
    loss_u, dl_u = K.ce_fwd(lg, t.to(DEV), 1.0 / R)
    report(f'linear_ce[{R},{C}].loss_vs_unfused', loss.reshape(1), loss_u.reshape(1), 2e-6)
    report(f'linear_ce[{R},{C}].dlogits_vs_unfused', dl.hi.float(), dl_u.hi.float(), 2 ** -7)
    loss_only, none = K.linear_ce(hb, wb, t.to(DEV), 1.0 / R, want_grad=False)
    assert none.hi is None and torch.equal(loss_only, loss)
    bad = t.clone()
    bad[5] = C
    assert torch.isnan(K.linear_ce(hb, wb, bad.to(DEV), 1.0 / R, want_grad=False)[0])
    assert K.linear_ce(K.BF(h.to(DEV), h.to(DEV)), wb, t.to(DEV), 1.0 / R) is None          # one pair, one plain operand: unfused path


@pytest.mark.parametrize('R,C,Kd', [(1000, 8192, 512), (2560, 512, 256), (300, 192, 64)])
def test_fused_linear_cross_entropy_hi_lo(R, C, Kd):
    """the same on the hi + lo ring (amdnuwa_linear_ce_x3: the to_logits of 'bf16x3-fwd', np.py:1958-1963): fp32 operands carried as
    bf16 pairs, loss / dlogits against torch in fp64 on the fp32 operands and against the unfused pair (x3 gemm_nt + ce_fwd);
    the all-pairs mode 'bf16x3' (dlogits wanted as a pair) keeps the unfused path"""
    from nuwa_pytorch_amd import kernels as K
    g = torch.Generator().manual_seed(R + C + 1)
    h = torch.randn(R, Kd, generator=g) * 0.8
    w = torch.randn(C, Kd, generator=g) * (3.0 / Kd ** 0.5)
    t = torch.randint(0, C, (R,), generator=g)
    t[0], t[1], t[-1] = 0, C - 1, C - 1
    logits = h.double() @ w.double().t()
    ref_loss = torch.nn.functional.cross_entropy(logits, t).float()
    ref_dl = ((logits.softmax(-1) - torch.nn.functional.one_hot(t, C).double()) / R).float()
    hb, wb = to_bf_pair(h.to(DEV), True), to_bf_pair(w.to(DEV), True)
    prev = K.get_precision()
    try:
        K.set_precision('bf16x3-fwd')
        out = K.linear_ce(hb, wb, t.to(DEV), 1.0 / R)
        assert out is not None
        loss, dl = out
        assert dl.lo is None
        report(f'linear_ce_x3[{R},{C}].loss', loss.reshape(1), ref_loss.reshape(1), 2e-6)
        report(f'linear_ce_x3[{R},{C}].dlogits', dl.hi.float(), ref_dl, 2 ** -8)
        lg = K.gemm_nt(hb, wb)
        report(f'linear_ce_x3[{R},{C}].logits_x3', lg, logits.float(), 2e-5)
        loss_u, dl_u = K.ce_fwd(lg, t.to(DEV), 1.0 / R, lo=False)
        report(f'linear_ce_x3[{R},{C}].loss_vs_unfused', loss.reshape(1), loss_u.reshape(1), 2e-6)
        report(f'linear_ce_x3[{R},{C}].dlogits_vs_unfused', dl.hi.float(), dl_u.hi.float(), 2 ** -7)
        loss_only, none = K.linear_ce(hb, wb, t.to(DEV), 1.0 / R, want_grad=False)
        assert none.hi is None and torch.equal(loss_only, loss)
        again = K.linear_ce(hb, wb, t.to(DEV), 1.0 / R)
        assert torch.equal(again[0], loss) and torch.equal(again[1].hi, dl.hi)               # fixed order: bit-repeatable
        # the dlogits pass on ONE fp16 MFMA per product (what 'bf16x3-fwd' trains with): same loss bit for bit (pass 1 is unchanged),
        # dlogits inside the same bound against fp64
        loss16, dl16 = K.linear_ce(hb, wb, t.to(DEV), 1.0 / R, w16=w.to(DEV).half().contiguous())
        assert torch.equal(loss16, loss)
        report(f'linear_ce_x3[{R},{C}].dlogits_f16_pass', dl16.hi.float(), ref_dl, 2 ** -8)
        bad = t.clone()
        bad[5] = C
        assert torch.isnan(K.linear_ce(hb, wb, bad.to(DEV), 1.0 / R, want_grad=False)[0])
        K.set_precision('bf16x3')
        assert K.linear_ce(hb, wb, t.to(DEV), 1.0 / R) is None
    finally:
        K.set_precision(prev)


# ---------------------------------------------------------------------------------------------------
# fp16 gradients of the 'bf16x3-fwd' backward (round 5): gradients travel as fp16(S * value), S a device-side power of two
# ---------------------------------------------------------------------------------------------------

def _s2(S):
    return torch.tensor([S, 1.0 / S], dtype=torch.float32, device=DEV)


@pytest.mark.parametrize('S', [1.0, 2.0 ** 18, 2.0 ** -6])
def test_layernorm_backward_with_fp16_gradients(K, S):
    """post-norm backward writing dy as fp16(S dy), pre-norm backward / the chained form reading dh = fp16(S dh): against the fp32
    kernels on the same inputs; the weight-gradient partials must not see the scale"""
    torch.manual_seed(3)
    R, D = 2560 + 7, 512
    gmag = 4.0 / S                                   # a gradient magnitude that S brings to O(1)
    x = torch.randn(R, D, device=DEV) * 2 + 0.5
    w = torch.randn(D, device=DEV)
    b = torch.randn(D, device=DEV)
    g = torch.randn(R, D, device=DEV) * gmag
    _, m, r, _ = K.ln_fwd(x, w, b)
    s2 = _s2(S)
    dx32, dw32, db32, _ = K.ln_bwd(g, x, m, r, w)                      # fp32 reference: dx = LN backward of g (dres None -> zeros)
    dy16, dw16, db16, ds16 = K.ln_bwd(g, x, m, r, w, to_f16=s2, want_dsum=True)
    assert isinstance(dy16, K.G16) and dy16.t.dtype == torch.float16
    report(f'ln_bwd.f16_out[S={S}]', dy16.t.float() / S, dx32, 2 ** -10)
    assert torch.equal(dw16, dw32) and torch.equal(db16, db32)
    report(f'ln_bwd.f16_out.dsum[S={S}]', ds16, dx32.sum(0), 1e-4)
    # pre-norm backward from an fp16 dh, accumulating into the fp32 stream gradient
    dh = torch.randn(R, D, device=DEV) * gmag
    dh16 = K.G16((dh * S).half(), s2)
    dhv = dh16.t.float() / S                                               # what the kernel sees
    ref, rw, rb, _ = K.ln_bwd(dhv, x, m, r, w, dres=g)
    got, gw, gb, _ = K.ln_bwd(dh16, x, m, r, w, dres=g)
    report(f'ln_bwd.f16_in[S={S}]', got, ref, 2e-6)
    report(f'ln_bwd.f16_in.dw[S={S}]', gw, rw, 2e-6)
    # chained: pre-norm backward of block k+1 (fp16 dh) + post-norm backward of block k (fp16 dy out)
    y = torch.randn(R, D, device=DEV)
    w2 = torch.randn(D, device=DEV)
    _, m2, r2, _ = K.ln_fwd(y, w2, b)
    dxr, dwr, dbr, dypr, dwpr, dbpr, dspr = K.ln_bwd_chain(dhv, x, m, r, w, g, y, m2, r2, w2, want_dsum=True)
    dxc, dwc, dbc, dypc, dwpc, dbpc, dspc = K.ln_bwd_chain(dh16, x, m, r, w, g, y, m2, r2, w2, want_dsum=True, out_f16=s2)
    report(f'ln_bwd_chain.f16.dx[S={S}]', dxc, dxr, 2e-6)
    report(f'ln_bwd_chain.f16.dy_prev[S={S}]', dypc.t.float() / S, bf_value(dypr) if dypr.lo is not None else dypr.hi.float(), 2 ** -7)
    report(f'ln_bwd_chain.f16.dw_prev[S={S}]', dwpc, dwpr, 2e-6)
    report(f'ln_bwd_chain.f16.dsum_prev[S={S}]', dspc, dspr, 2e-6)
    assert K.f16_sat_count() == 0


def test_fp16_gradient_gemms(K):
    """the four products of the FeedForward backward on fp16 operands: dgg (+ the gate's backward in the epilogue, fp16 du), dh (fp16 out),
    and the two weight gradients (TN, alpha / S from the device scalar) -- against fp64 on the same fp16 operand values"""
    torch.manual_seed(5)
    R, D, FP, FFI, S = 2560 * 4, 512, 1376, 1365, 2.0 ** 10
    s2 = _s2(S)
    dy = (torch.randn(R, D, device=DEV) * 0.01 * S).half()                 # = S * dy
    w2T = (torch.randn(FP, D, device=DEV) * 0.2).half()
    u = (torch.randn(R, 2 * FP, device=DEV)).to(torch.bfloat16)             # interleaved layout
    assert K.gemm_nt_f16ops_ok(R, FP, D, out_bf16=True, geglu_bwd=True) and K.gemm_nt_f16ops_ok(R, D, 2 * FP, out_bf16=False, out_f16=True)
    du = K.gemm_nt_geglu_bwd16(dy, w2T, u, FP)
    assert du.dtype == torch.float16 and du.shape == (R, 2 * FP)
    dgg = dy.double() @ w2T.double().t()
    ud = K.geglu_deinterleave(u.double().cpu(), FP, dim=1).requires_grad_(True)
    (ud[:, :FP] * F.gelu(ud[:, FP:])).backward(dgg.cpu())
    report('bwd16.du', K.geglu_deinterleave(du.float().cpu(), FP, dim=1), ud.grad.float(), 2 ** -9)
    w1T = (torch.randn(D, 2 * FP, device=DEV) * 0.1).half()
    dh = K.gemm_nt_f16ops(du, w1T, out_f16=True)
    report('bwd16.dh', dh.float(), (du.double() @ w1T.double().t()).float(), 2 ** -10)
    # weight gradients: the 1 / S comes from the device scalar
    gg = (torch.randn(R, FP, device=DEV)).half()
    assert K.gemm_tn16_ok(R, D, FFI, lda=D, ldb=FP) and K.gemm_tn16_ok(R, 2 * FP, D)
    dw2 = torch.empty(D, FFI, device=DEV)
    K.gemm_tn16(dy, gg, dw2, s2, N2=FFI)
    report('bwd16.dw2', dw2, ((dy.double().t() @ gg.double())[:, :FFI] / S).float(), 1e-5)
    h = torch.randn(R, D, device=DEV).half()
    dw1 = torch.empty(2 * FP, D, device=DEV)
    K.gemm_tn16(du, h, dw1, s2)
    report('bwd16.dw1', dw1, (du.double().t() @ h.double() / S).float(), 1e-5)
    assert not K.gemm_tn16_ok(321, 512, 512)                                 # odd token counts stay on the bf16 backward
    # saturation is DEFINED and counted
    K.f16_sat_count()
    big = (torch.ones(256 * 5, D, device=DEV) * 300).half()
    out = K.gemm_nt_f16ops(big, (torch.ones(D, D, device=DEV)).half(), out_f16=True)        # 300 * 512 > 65504
    assert torch.isfinite(out).all() and float(out.max()) == 65504.0
    assert K.f16_sat_count() > 0 and K.f16_sat_count() == 0


def test_fp16_store_keeps_nan(K):
    """advisor (round 4): the saturating fp16 stores must not turn a NaN activation into -65504"""
    x = torch.randn(8, 512, device=DEV)
    x[3, 7] = float('nan')
    h, _, _, _ = K.ln_fwd(x, torch.ones(512, device=DEV), torch.zeros(512, device=DEV), f16=True)
    assert torch.isnan(h.f16[3]).all() and torch.isnan(h.hi[3]).all()
    assert torch.isfinite(h.f16[2]).all()


def test_every_kernel_family_bit_reproducible_with_coresident_workgroups(K):
    """round 5: the reproducibility stress of round 4 covered the attention kernels; the GEMM epilogues (fp16 forward forms, two-MFMA form,
    GEGLU backward in bf16 and in fp16), the weight gradients, the LayerNorm kernels and the cross entropy share the source patterns of the
    defect it found, so they take the same test: 5 runs on the same inputs at a batch that fills every CU several times over (b = 32),
    element for element (tools/determinism_stress.py; the evidence run of the round is b = 128 x 10: profiles/r05_determinism_stress.txt)"""
    import os
    import sys
    from gpu_util import ROOT
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import determinism_stress as DS
    DS.REP = 5
    bad = []

    def check(name, fn):
        n = DS.check(name, fn)
        if n:
            bad.append((name, n))
        return n
    assert DS.other_families(32, check) == 0, bad


def test_attention_cores_at_batch_16_equal_their_one_sample_results(K):
    """round 5 (review): reproducibility says 'the same wrong value every time' is fine; this does not.  The attention cores work per sample,
    so sample s of a batch-16 launch -- two workgroups per CU, several rounds -- must equal the batch-1 launch on that sample's data BIT FOR
    BIT (forward outputs, statistics, dq / dk / dv, dS / P'): a defect that needs co-resident workgroups shows up as a difference."""
    heads, dh, B, n, T = 8, 64, 16, 2560, 256
    inner = heads * dh
    torch.manual_seed(2)
    wth = (torch.randn(heads, heads) * 0.5 + torch.eye(heads)).to(DEV)
    pick = (0, 7, 15)
    for dil in ((1, 1, 1), (2, 2, 2), (4, 4, 4)):
        qkv = torch.randn(B * n, 3 * inner, device=DEV)
        dO = torch.randn(B * n, inner, device=DEV).to(torch.bfloat16)
        g = K.s3_geom(B, n, (10, 16, 16), (5, 3, 3), dil, heads, dh)
        g1 = K.s3_geom(1, n, (10, 16, 16), (5, 3, 3), dil, heads, dh)
        p16 = K.BF(qkv.to(torch.bfloat16), None, qkv.half())
        o = K.sparse3dna_fwd(g, p16, wth, o_f16=True)
        dqkv, _, _ = K.sparse3dna_bwd(g, K.BF(p16.hi, None), wth, K.BF(dO, None))
        for s in pick:
            rows = slice(s * n, (s + 1) * n)
            p1 = K.BF(p16.hi[rows].contiguous(), None, p16.f16[rows].contiguous())
            o1 = K.sparse3dna_fwd(g1, p1, wth, o_f16=True)
            assert torch.equal(o.hi[rows], o1.hi) and torch.equal(o.f16[rows], o1.f16), f'3DNA forward, dilation {dil[0]}, sample {s}'
            d1, _, _ = K.sparse3dna_bwd(g1, K.BF(p1.hi, None), wth, K.BF(dO[rows].contiguous(), None))
            assert torch.equal(dqkv.hi[rows], d1.hi), f'3DNA backward, dilation {dil[0]}, sample {s}'
    q, kv = torch.randn(B * n, inner, device=DEV), torch.randn(B * T, 2 * inner, device=DEV)
    mask = (torch.rand(B, T, device=DEV) > 0.2).to(torch.uint8)
    nk, nv = torch.randn(heads, dh, device=DEV), torch.randn(heads, dh, device=DEV)
    dO = torch.randn(B * n, inner, device=DEV).to(torch.bfloat16)
    gx, gx1 = K.x_geom(B, n, T, heads, dh), K.x_geom(1, n, T, heads, dh)
    q16, kv16 = K.BF(q.to(torch.bfloat16), None, q.half()), K.BF(kv.to(torch.bfloat16), None, kv.half())
    pk = K.xattn_pack(gx, kv16, nk, nv, mask)
    o, st = K.xattn2_fwd_f16(gx, q16, pk, wth, o_f16=True)
    pkb = K.xattn_pack(gx, K.BF(kv16.hi, None), nk, nv, mask)
    dq, dS, Pm, _ = K.xattn2_bwd(gx, K.BF(q16.hi, None), K.BF(dO, None), pkb, wth, st)
    for s in pick:
        rows, krows = slice(s * n, (s + 1) * n), slice(s * T, (s + 1) * T)
        q1 = K.BF(q16.hi[rows].contiguous(), None, q16.f16[rows].contiguous())
        kv1 = K.BF(kv16.hi[krows].contiguous(), None, kv16.f16[krows].contiguous())
        pk1 = K.xattn_pack(gx1, kv1, nk, nv, mask[s:s + 1].contiguous())
        o1, st1 = K.xattn2_fwd_f16(gx1, q1, pk1, wth, o_f16=True)
        assert torch.equal(o.hi[rows], o1.hi) and torch.equal(o.f16[rows], o1.f16), f'cross attention forward, sample {s}'
        assert torch.equal(st.reshape(B, -1)[s], st1.reshape(-1)), f'cross attention statistics, sample {s}'
        pkb1 = K.xattn_pack(gx1, K.BF(kv1.hi, None), nk, nv, mask[s:s + 1].contiguous())
        dq1, dS1, Pm1, _ = K.xattn2_bwd(gx1, K.BF(q1.hi, None), K.BF(dO[rows].contiguous(), None), pkb1, wth, st1)
        # (chunk-major arrays: the columns that hold a key -- the padding lane groups of the last chunk write nothing)
        assert torch.equal(dq.hi[rows], dq1.hi) and torch.equal(K.xattn_rows(gx, dS.hi)[s], K.xattn_rows(gx1, dS1.hi)[0]) and \
            torch.equal(K.xattn_rows(gx, Pm.hi)[s], K.xattn_rows(gx1, Pm1.hi)[0]), f'cross attention backward, sample {s}'


@pytest.mark.parametrize('dil', [1, 2, 4])
def test_sparse3dna_bwd_packed_workspace_equals_the_fp32_workspace(K, dil):
    """round 5: the MFMA backward hands ds / P' from the query side to the key side as ONE array of (bf16 ds | bf16 P') words instead of two
    fp32 arrays.  The key side rounded both to bf16 for its MFMA operands anyway: dq, dk, dv and dW_th must not change by a bit
    (tuning key 24 = 1 selects the fp32 pair); partial last rows included (n = 2000)."""
    from nuwa_pytorch_amd import _lib
    L = _lib.lib()
    heads, dh, B = 8, 64, 3
    inner = heads * dh
    torch.manual_seed(dil)
    wth = (torch.randn(heads, heads) * 0.5 + torch.eye(heads)).to(DEV)
    for n in (2560, 2000):
        qkv = K.BF(torch.randn(B * n, 3 * inner, device=DEV).to(torch.bfloat16), None)
        dO = K.BF(torch.randn(B * n, inner, device=DEV).to(torch.bfloat16), None)
        g = K.s3_geom(B, n, (10, 16, 16), (5, 3, 3), (dil, dil, dil), heads, dh)
        try:
            L.amdnuwa_set_tuning(24, 1)
            d_ref, w_ref, _ = K.sparse3dna_bwd(g, qkv, wth, dO)
            L.amdnuwa_set_tuning(24, 0)
            d_new, w_new, _ = K.sparse3dna_bwd(g, qkv, wth, dO)
        finally:
            L.amdnuwa_set_tuning(24, 0)
        assert torch.equal(d_ref.hi, d_new.hi), f'dqkv differs (dilation {dil}, n = {n})'
        assert torch.equal(w_ref, w_new), f'dW_th differs (dilation {dil}, n = {n})'
