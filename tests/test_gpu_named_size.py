"""Parity at the sizes BASELINE.json NAMES (not fixture sizes), on the MI355X against the oracle on the same seeded inputs.

  cfg 3  dim 512, 8 heads x 64, 10x16x16 tokens, 3DNA kernel (5,3,3), dilation cycle (1,2,4), 256 text tokens, codebook 8192
         (i)  one decoder layer per dilation, forward + backward (every parameter gradient, dx, dcontext)
         (ii) the full 24-layer decoder: logits of one sample vs the oracle in BOTH precision modes; the measured errors go to
              gpurun_out/parity_log.jsonl and gpurun_out/named_size.json (bench.py quotes them)
  cfg 2  dim 256, FFI 682, 4x16x16 tokens, kernel (3,3,3): one decoder layer fwd + bwd (K = 256 GEMMs, odd FFI padding)
  cfg 4  dim 512 reversible: two reversible depths fwd + bwd through the recomputing backward vs the oracle's plain form;
         a depth-64 training step (finite loss / gradients, activation memory flat in depth)
  cfg 5  dim 512 dual decoder: one depth (video triple, audio triple, cross-modality pair) fwd + bwd vs the oracle

Tolerances (max-abs error / max-abs reference), as everywhere in tests/:
  'bf16x3'     parity mode     : outputs 1e-3 (the north-star bound), gradients 2e-3
  'bf16x3-fwd' compliant mode  : outputs 1e-3 (3-MFMA hi + lo projections, attention cores and FeedForward GEMMs on single fp16
                                 MFMAs; with the fp16 parts switched off its forward IS the bf16x3 forward: asserted bit-identical),
                                 gradients as 'bf16' (its backward runs single bf16 MFMAs on the hi parts)
  'bf16'       fast mode       : outputs 9e-3, gradients 1.4e-2 = 1.5 x the largest errors measured at these sizes in round 2
                                 (5.6e-3 / 9.1e-3, profiles/r02h_named_size.json); full-depth logits 1.2e-2 (measured 7.9e-3)
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import report, record, rel_err, rel_l2, ROOT  # noqa: E402

DEV = 'cuda'
MODES = [('bf16x3', 1e-3, 2e-3), ('bf16x3-fwd', 1e-3, 1.4e-2), ('bf16', 9e-3, 1.4e-2)]
SUMMARY = os.path.join(ROOT, 'gpurun_out', 'named_size.json')


@pytest.fixture(scope='module')
def A():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import nuwa_pytorch_amd
    return nuwa_pytorch_amd


@pytest.fixture(scope='module')
def O():
    from oracle import nuwa_oracle
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    return nuwa_oracle


def _note(key, value):
    try:
        os.makedirs(os.path.dirname(SUMMARY), exist_ok=True)
        d = {}
        if os.path.exists(SUMMARY):
            with open(SUMMARY) as f:
                d = json.load(f)
        d[key] = value
        with open(SUMMARY, 'w') as f:
            json.dump(d, f, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass


def _cpu_params(mod):
    return {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()}


def _req(P):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in P.items()}


def _layer_case(A, O, *, dim, video_shape, kernel, dil, T, b, tag):
    """one Transformer decoder layer (3DNA + text cross-attention + FF, sandwich norms, token shift) at a named size:
    product (both modes) vs oracle.decoder_layer, forward and backward"""
    import nuwa_pytorch_amd.nuwa_pytorch as M
    torch.manual_seed(0)
    tr = M.Transformer(dim=dim, depth=1, causal=True, heads=8, dim_head=64, cross_attend=True, sparse_3dna_attn=True,
                       sparse_3dna_kernel_size=kernel, sparse_3dna_video_shape=video_shape, sparse_3dna_dilations=(dil,),
                       shift_video_tokens=True)
    with torch.no_grad():                      # non-trivial norm parameters and biases
        for n_, p in tr.named_parameters():
            if 'norm' in n_ or n_.endswith('.bias'):
                p.add_(0.1 * torch.randn_like(p))
    P = _cpu_params(tr)
    n = video_shape[0] * video_shape[1] * video_shape[2]
    g = torch.Generator().manual_seed(7)
    x = torch.randn(b, n, dim, generator=g)
    ctx = torch.randn(b, T, dim, generator=g)
    mask = torch.ones(b, T, dtype=torch.bool)
    mask[:, -T // 4:] = torch.rand(b, T // 4, generator=g) > 0.5
    dy = torch.randn(b, n, dim, generator=g)
    cfg = dict(video_shape=video_shape, kernel_size=kernel, dilations=(dil,), heads=8, depth=1, shift=True)
    Pr = _req(P)
    xr, cr = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    yr = O.decoder_layer(xr, O.sub(Pr, 'layers.0'), cfg, 0, cr, mask)
    yr.backward(dy)
    tr = tr.to(DEV)
    out = {}
    for mode, tol, gtol in MODES:
        A.set_precision(mode)
        try:
            tr.zero_grad(set_to_none=True)
            xd, cd = x.to(DEV).requires_grad_(True), ctx.to(DEV).requires_grad_(True)
            y = tr.forward_layers(xd, context=cd, context_mask=mask.to(DEV))
            e = {'y': report(f'{tag}[{mode}].y', y, yr.detach(), tol)}
            y.backward(dy.to(DEV))
            e['dx'] = report(f'{tag}[{mode}].dx', xd.grad, xr.grad, gtol)
            e['dctx'] = report(f'{tag}[{mode}].dctx', cd.grad, cr.grad, gtol)
            worst = 0.
            for k, p in tr.named_parameters():
                gr = Pr[k].grad
                if gr is None:                  # the stack's final StableLayerNorm is not part of forward_layers
                    assert k.startswith('norm.'), k
                    continue
                assert p.grad is not None, k
                worst = max(worst, report(f'{tag}[{mode}].grad.{k}', p.grad, gr, gtol))
            e['worst_param_grad'] = worst
            out[mode] = e
        finally:
            A.set_precision('bf16')
    _note(tag, out)


@pytest.mark.parametrize('dil', [1, 2, 4])
def test_cfg3_decoder_layer_vs_oracle(A, O, dil):
    _layer_case(A, O, dim=512, video_shape=(10, 16, 16), kernel=(5, 3, 3), dil=dil, T=256, b=1, tag=f'cfg3.layer.dil{dil}')


def test_cfg2_decoder_layer_vs_oracle(A, O):
    # dim 256 -> FFI 682 (padded to 704 inside the bf16 copies), K = 256 GEMMs, 4 frames, cubic kernel 3
    _layer_case(A, O, dim=256, video_shape=(4, 16, 16), kernel=(3, 3, 3), dil=1, T=256, b=2, tag='cfg2.layer')


def test_cfg3_full_depth_logits_vs_oracle(A, O):
    """the whole named decoder (24 layers, dim 512, n = 2560, codebook 8192), one sample: logits vs the oracle.
    'bf16x3' and 'bf16x3-fwd' (same forward, bit for bit) must meet the north-star 1e-3; the fast 'bf16' mode's error is MEASURED
    here and bounded by 1.5 x its round-2 value."""
    import bench
    c = bench.CFGS['cfg3']
    torch.manual_seed(0)
    nuwa = bench.build_model(c, 'cpu')
    P = {k: v.detach().clone() for k, v in nuwa.state_dict().items() if not k.startswith('vae.') and not k.startswith('text_')}
    N = c['frames'] * c['fmap'] ** 2
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, c['codebook'], (1, N), generator=g)
    ctx = torch.randn(1, c['text_len'], c['dim'], generator=g)
    mask = torch.ones(1, c['text_len'], dtype=torch.bool)
    mask[:, -64:] = torch.rand(1, 64, generator=g) > 0.5
    cfg = dict(video_shape=(c['frames'], c['fmap'], c['fmap']), kernel_size=c['kernel'], dilations=c['dilation'], heads=c['heads'],
               depth=c['dec_depth'], shift=True)
    with torch.no_grad():
        loss_r, logits_r = O.decoder_loss(P, cfg, ids, ctx, mask, training=True, return_logits=True)
    nuwa = nuwa.to(DEV).train()
    res, got = {}, {}
    for mode, tol, _ in MODES:
        A.set_precision(mode)
        try:
            with torch.no_grad():
                x = nuwa.embed_video(ids.to(DEV)[:, :-1])
                h = nuwa.decode_hidden(x, ctx.to(DEV), mask.to(DEV))
                logits = nuwa._final(h)
                loss = nuwa._final(h, ids.to(DEV))
            res[mode] = dict(logits_rel_max=rel_err(logits, logits_r), logits_rel_l2=rel_l2(logits, logits_r),
                             loss=float(loss), loss_ref=float(loss_r), loss_rel=abs(float(loss) - float(loss_r)) / abs(float(loss_r)))
            got[mode] = logits.float().cpu()
        finally:
            A.set_precision('bf16')
    _note('cfg3.full_depth_logits', res)
    for mode, tol, _ in MODES:
        record(f'cfg3.full24[{mode}].logits', res[mode]['logits_rel_max'], res[mode]['logits_rel_l2'], tol)
    assert res['bf16x3']['logits_rel_max'] <= 1e-3, res
    assert res['bf16x3']['loss_rel'] <= 1e-4, res
    # with its fp16 attention cores and fp16 FeedForward GEMMs switched off the compliant mode runs the bf16x3 forward bit for bit
    from nuwa_pytorch_amd import kernels as KK
    KK.set_cores_f16(False)
    KK.set_ff_f16(False)
    KK.set_proj_f16x2(False)
    A.set_precision('bf16x3-fwd')
    try:
        with torch.no_grad():
            h = nuwa.decode_hidden(nuwa.embed_video(ids.to(DEV)[:, :-1]), ctx.to(DEV), mask.to(DEV))
            same = nuwa._final(h).float().cpu()
        # ... and what each class of two-MFMA products (fp16 activation x fp16 hi + lo weight) costs on top of the one-MFMA parts
        KK.set_cores_f16(True)
        KK.set_ff_f16(True)
        for cls in ('', 'o', 'oq', 'oql'):
            KK.set_proj_f16x2(cls)
            with torch.no_grad():
                h = nuwa.decode_hidden(nuwa.embed_video(ids.to(DEV)[:, :-1]), ctx.to(DEV), mask.to(DEV))
                lg = nuwa._final(h)
            res[f"bf16x3-fwd (two-MFMA classes '{cls}')"] = dict(logits_rel_max=rel_err(lg, logits_r), logits_rel_l2=rel_l2(lg, logits_r))
    finally:
        KK.set_cores_f16(True)
        KK.set_ff_f16(True)
        KK.set_proj_f16x2(os.environ.get('AMDNUWA_F16X2', KK.DEFAULT_F16X2))
        A.set_precision('bf16')
    assert torch.equal(same, got['bf16x3']), 'bf16x3-fwd without its fp16 parts must be the bf16x3 forward'
    res['bf16x3-fwd (all 3-MFMA)'] = dict(logits_rel_max=rel_err(same, logits_r))
    assert all(v['logits_rel_max'] <= 1e-3 for k, v in res.items() if k.startswith('bf16x3-fwd (two-MFMA')), res
    _note('cfg3.full_depth_logits', res)
    assert res['bf16x3-fwd']['logits_rel_max'] <= 1e-3, res
    assert res['bf16']['logits_rel_max'] <= 1.2e-2, res
    assert res['bf16']['loss_rel'] <= 2e-5, res


@pytest.mark.parametrize('mode', ['bf16x3-fwd', 'bf16'])
def test_cfg3_decoder_step_is_bit_reproducible_at_a_chip_filling_batch(A, mode):
    """the bench step (embedding -> 3 decoder layers with dilations 1, 2, 4 -> logits -> cross entropy -> backward) at b = 16: every kernel of the
    path runs with several workgroups per CU and several rounds of them.  Twice from the same inputs: the loss and every gradient bit for
    bit (no atomics on the path, fixed-order split-K sums).  The one-sample tests cannot see a defect that needs co-resident workgroups."""
    import bench
    c = dict(bench.CFGS['cfg3'], dec_depth=3)
    nuwa = bench.build_model(c, DEV)
    params = bench.decoder_params(nuwa)
    g = torch.Generator().manual_seed(5)
    b, N = 16, c['frames'] * c['fmap'] ** 2
    ids = torch.randint(0, c['codebook'], (b, N), generator=g).to(DEV)
    ctx = torch.randn(b, c['text_len'], c['dim'], generator=g).to(DEV)
    mask = (torch.rand(b, c['text_len'], generator=g) > 0.1).to(DEV)
    A.set_precision(mode)
    try:
        runs = []
        for _ in range(3):
            for p in params:
                p.grad = None
            loss = bench.decoder_step(nuwa, ids, ctx, mask)
            torch.cuda.synchronize()
            runs.append((None if loss is None else loss.detach().clone(), [p.grad.detach().clone() for p in params]))
    finally:
        A.set_precision('bf16')
    for k in (1, 2):
        if runs[0][0] is not None:
            assert torch.equal(runs[0][0], runs[k][0]), f'{mode}: loss differs between runs'
        bad = [i for i, (x, y) in enumerate(zip(runs[0][1], runs[k][1])) if not torch.equal(x, y)]
        assert not bad, f'{mode}: {len(bad)} of {len(params)} gradients differ between two runs of the same step'


def test_cfg4_reversible_blocks_vs_oracle(A, O):
    """cfg 4 (dec_reversible=True) at dim 512 / n = 2560: two reversible depths (3DNA|FF, cross|FF, twice) through the recomputing
    backward against the oracle's plain (stored-activation) evaluation of the same arithmetic"""
    import nuwa_pytorch_amd.nuwa_pytorch as M
    vs, kernel = (10, 16, 16), (5, 3, 3)
    torch.manual_seed(0)
    tr = M.ReversibleTransformer(dim=512, depth=2, causal=True, heads=8, dim_head=64, cross_attend=True, sparse_3dna_attn=True,
                                 sparse_3dna_kernel_size=kernel, sparse_3dna_video_shape=vs, sparse_3dna_dilations=(1, 2, 4),
                                 shift_video_tokens=True)
    P = {k: v for k, v in _cpu_params(tr).items() if not k.startswith('net.')}
    g = torch.Generator().manual_seed(5)
    n = 2560
    x, ctx, dy = torch.randn(1, n, 512, generator=g), torch.randn(1, 256, 512, generator=g), torch.randn(1, n, 512, generator=g)
    mask = torch.ones(1, 256, dtype=torch.bool)
    mask[:, 200:] = False
    cfg = dict(video_shape=vs, kernel_size=kernel, dilations=(1, 2, 4), heads=8, depth=2, shift=True)
    Pr = _req(P)
    xr, cr = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    yr = O.reversible_decoder_stack(xr, Pr, cfg, cr, mask)
    yr.backward(dy)
    tr = tr.to(DEV).train()
    # 'bf16x3' (parity mode) and 'bf16x3-fwd' (the benchmarked mode: the recomputing backward re-enters its forward arithmetic, the
    # gradient products run single bf16 MFMAs)
    for mode, tol, gtol, ptol in (('bf16x3', 1e-3, 2e-3, 3e-3), ('bf16x3-fwd', 1e-3, 2e-2, 2e-2)):
        A.set_precision(mode)
        try:
            tr.zero_grad(set_to_none=True)
            xd, cd = x.to(DEV).requires_grad_(True), ctx.to(DEV).requires_grad_(True)
            y = tr(xd, context=cd, context_mask=mask.to(DEV))
            res = {'y': report(f'cfg4.rev2[{mode}].y', y, yr.detach(), tol)}
            y.backward(dy.to(DEV))
            res['dx'] = report(f'cfg4.rev2[{mode}].dx', xd.grad, xr.grad, gtol)
            res['dctx'] = report(f'cfg4.rev2[{mode}].dctx', cd.grad, cr.grad, gtol)
            first = 'layers.0.0.fn.fn.to_q.weight'
            last = 'layers.3.1.fn.fn.net.3.weight'
            named = dict(tr.named_parameters())
            res['worst_param_grad'] = max(report(f'cfg4.rev2[{mode}].grad.{k}', named[k].grad, Pr[k].grad, ptol)
                                          for k in (first, 'layers.1.0.fn.to_kv.weight', 'layers.2.0.fn.fn.to_out.weight', last, 'norm.norm.weight'))
            _note(f'cfg4.rev2[{mode}]', res)
        finally:
            A.set_precision('bf16')


def test_cfg4_depth64_step_is_finite_and_memory_flat(A):
    """the named cfg 4 (depth 64, reversible, dim 512, 10x16x16): one training step is finite and the activation memory does
    not grow with depth (depth 64 vs depth 8 through the recomputing backward)"""
    import bench
    peaks = {}
    for depth in (8, 64):
        c = dict(bench.CFGS['cfg4'], dec_depth=depth)
        nuwa = bench.build_model(c, DEV)
        params = bench.decoder_params(nuwa)
        for p in nuwa.parameters():
            p.requires_grad_(False)
        for p in params:
            p.requires_grad_(True)
        g = torch.Generator().manual_seed(3)
        N = 2560
        ids = torch.randint(0, c['codebook'], (2, N), generator=g).to(DEV)
        ctx = torch.randn(2, 256, 512, generator=g).to(DEV)
        mask = torch.ones(2, 256, dtype=torch.bool, device=DEV)
        bench.decoder_step(nuwa, ids, ctx, mask)           # builds the bf16 weight copies, gradients
        torch.cuda.synchronize()
        for p in params:
            p.grad = None
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss = bench.decoder_step(nuwa, ids, ctx, mask)
        torch.cuda.synchronize()
        grads_bytes = sum(p.grad.numel() * 4 for p in params if p.grad is not None)
        peaks[depth] = (torch.cuda.max_memory_allocated() - base - grads_bytes) / 2 ** 20
        assert torch.isfinite(loss.detach()), float(loss.detach())
        assert all(torch.isfinite(p.grad).all() for p in params if p.grad is not None)
        assert abs(float(loss.detach()) - 9.2) < 1.0, float(loss.detach())        # ~ln(8192) + small at random init
        del nuwa, params, loss
        torch.cuda.empty_cache()
    _note('cfg4.activation_mb', peaks)
    assert peaks[64] < 1.35 * peaks[8] + 64, peaks


def test_cfg5_dual_decoder_layer_vs_oracle(A, O):
    """cfg 5 at its named width (dim 512, 10x16x16 video tokens + 320 audio tokens, 32 per frame): ONE depth of the dual decoder
    with its cross-modality pair (video / audio triples, one-frame-lagged video<->audio attention, FFs) fwd + bwd vs the oracle"""
    from nuwa_pytorch_amd.video_audio import DualModalityDecoder
    vs, kernel = (10, 16, 16), (5, 3, 3)
    torch.manual_seed(0)
    dec = DualModalityDecoder(dim=512, depth=1, num_audio_tokens_per_video_frame=32, num_video_tokens_per_frame=256,
                              sparse_3dna_video_shape=vs, heads=8, dim_head=64, sparse_3dna_kernel_size=kernel,
                              sparse_3dna_dilations=(2,), sparse_2dna_kernel_size=7, sparse_2dna_dilation=(2,),
                              sparse_2dna_rel_pos_bias=True, shift_video_tokens=True, shift_audio_tokens=True,
                              cross_modality_attn_every=1)
    P = _cpu_params(dec)
    g = torch.Generator().manual_seed(9)
    v = torch.randn(1, 2560, 512, generator=g)
    a = torch.randn(1, 321, 512, generator=g)
    ctx = torch.randn(1, 256, 512, generator=g)
    mask = torch.ones(1, 256, dtype=torch.bool)
    mask[:, 230:] = False
    dv, da = torch.randn(1, 2560, 512, generator=g), torch.randn(1, 321, 512, generator=g)
    cfg = dict(depth=1, heads=8, video_shape=vs, kernel_size=kernel, dilations=(2,), audio_kernel=7, audio_dilations=(2,), every=1,
               v_per_frame=256, a_per_frame=32, shift_video=True, shift_audio=True)
    Pr = _req(P)
    vr, ar, cr = (t.clone().requires_grad_(True) for t in (v, a, ctx))
    yv, ya = O.dual_decoder(vr, ar, Pr, cfg, cr, mask)
    torch.autograd.backward([yv, ya], [dv, da])
    dec = dec.to(DEV).train()
    for mode, tol, gtol, ptol in (('bf16x3', 1e-3, 2e-3, 3e-3), ('bf16x3-fwd', 1e-3, 2e-2, 2e-2)):
        A.set_precision(mode)
        try:
            dec.zero_grad(set_to_none=True)
            vd, ad, cd = (t.to(DEV).requires_grad_(True) for t in (v, a, ctx))
            ov, oa = dec(vd, ad, context=cd, context_mask=mask.to(DEV))
            res = {'video': report(f'cfg5.dual1[{mode}].video', ov, yv.detach(), tol), 'audio': report(f'cfg5.dual1[{mode}].audio', oa, ya.detach(), tol)}
            torch.autograd.backward([ov, oa], [dv.to(DEV), da.to(DEV)])
            res['dvideo'] = report(f'cfg5.dual1[{mode}].dvideo', vd.grad, vr.grad, gtol)
            res['daudio'] = report(f'cfg5.dual1[{mode}].daudio', ad.grad, ar.grad, gtol)
            res['dctx'] = report(f'cfg5.dual1[{mode}].dctx', cd.grad, cr.grad, gtol)
            worst = 0.
            for k, p in dec.named_parameters():
                if Pr[k].grad is None:
                    continue
                worst = max(worst, report(f'cfg5.dual1[{mode}].grad.{k}', p.grad, Pr[k].grad, ptol))
            res['worst_param_grad'] = worst
            _note(f'cfg5.dual1[{mode}]', res)
        finally:
            A.set_precision('bf16')


def test_fp16_range_guard_of_the_compliant_mode(A):
    """'bf16x3-fwd' runs the FeedForward / 3DNA q-k-v products on fp16 copies (5-bit exponent).  Defined behaviour at the edges:
    (i) activations saturate at +-65504 in every fp16 store (LayerNorm copy, GEMM epilogue copies) instead of becoming inf;
    (ii) a weight tensor whose largest magnitude leaves [2^-10, 6e4] is NOT converted: the block runs the bf16 hi + lo (3-MFMA)
    products, i.e. exactly the 'bf16x3' forward -- asserted bit for bit against that mode."""
    import nuwa_pytorch_amd.nuwa_pytorch as M
    from nuwa_pytorch_amd import kernels as K, ops
    # (i) LayerNorm fp16 copy and the fp16-operand GEMM epilogue copies saturate
    torch.manual_seed(0)
    R, D = 16384 + 8, 512
    x = torch.randn(R, D, device=DEV)
    w = torch.full((D,), 4.0e4, device=DEV)                    # |LN(x) * w| reaches ~1.6e5
    h, _, _, _ = K.ln_fwd(x, w, torch.zeros(D, device=DEV), f16=True)
    ref = torch.nn.functional.layer_norm(x, (D,), w, None).clamp(-65504, 65504)
    assert bool(torch.isfinite(h.f16.float()).all()) and float(h.f16.float().abs().max()) == 65504.0
    report('f16_guard.ln_saturates', h.f16.float(), ref, 2 ** -10)
    a16 = (torch.randn(R, D, device=DEV) * 30).half()
    b16 = (torch.randn(1536, D, device=DEV) * 30).half()        # products up to ~1e5
    out = K.gemm_nt_f16ops(a16, b16, out_bf16=True, copy_f16=True)
    exact = (a16.double() @ b16.double().t())
    assert float(exact.abs().max()) > 65504 and bool(torch.isfinite(out.f16.float()).all())
    report('f16_guard.gemm_copy_saturates', out.f16.float(), exact.clamp(-65504, 65504).float(), 2 ** -10)
    # (ii) weights outside the range: hi + lo products instead
    assert ops.f16_weights_ok(torch.randn(64, 64, device=DEV)) and not ops.f16_weights_ok(torch.randn(64, 64, device=DEV) * 1e5) \
        and not ops.f16_weights_ok(torch.randn(64, 64, device=DEV) * 1e-5)
    torch.manual_seed(1)
    tr = M.Transformer(dim=512, depth=1, causal=True, heads=8, dim_head=64, cross_attend=True, sparse_3dna_attn=True,
                       sparse_3dna_kernel_size=(5, 3, 3), sparse_3dna_video_shape=(10, 16, 16), sparse_3dna_dilations=(2,),
                       shift_video_tokens=False).to(DEV)
    g = torch.Generator().manual_seed(2)
    b = 8
    x = torch.randn(b, 2560, 512, generator=g).to(DEV)
    ctx = torch.randn(b, 256, 512, generator=g).to(DEV)
    mask = torch.ones(b, 256, dtype=torch.bool, device=DEV)
    ff = tr.layers[0][2].fn
    for case, scale in (('huge FeedForward weights', 2.0e5), ('tiny FeedForward weights', 1.0e-6)):
        with torch.no_grad():
            saved = [ff.net[0].weight.clone(), ff.net[3].weight.clone()]
            ff.net[0].weight.mul_(scale / float(ff.net[0].weight.abs().max()))
            ff.net[3].weight.mul_(1.0 / scale if scale > 1 else 1.0)          # keep the block's output O(1)
        outs = {}
        for mode in ('bf16x3-fwd', 'bf16x3'):
            A.set_precision(mode)
            K.set_cores_f16(False)                             # isolate the GEMM forms: attention cores on the hi + lo kernels in both runs
            try:
                with torch.no_grad():
                    outs[mode] = tr.forward_layers(x, context=ctx, context_mask=mask).float().cpu()
            finally:
                K.set_cores_f16(True)
                A.set_precision('bf16')
        with torch.no_grad():
            ff.net[0].weight.copy_(saved[0]); ff.net[3].weight.copy_(saved[1])
        assert bool(torch.isfinite(outs['bf16x3-fwd']).all()), case
        assert torch.equal(outs['bf16x3-fwd'], outs['bf16x3']), f'{case}: the guarded blocks must run the hi + lo products'
    # (iii) the two-MFMA products (fp16 activation x fp16 hi + lo weight): a to_out / to_q weight outside the range keeps its three-MFMA
    # product -- the run equals the one with the two-MFMA switch off, bit for bit, while an in-range stack differs from it
    s3, xa = tr.layers[0][0].fn, tr.layers[0][1].fn
    A.set_precision('bf16x3-fwd')
    try:
        def run(x2):
            K.set_proj_f16x2(x2)
            with torch.no_grad():
                return tr.forward_layers(x, context=ctx, context_mask=mask).float().cpu()
        assert not torch.equal(run('oq'), run(False))
        with torch.no_grad():
            saved = [s3.to_out.weight.clone(), xa.to_q.weight.clone()]
            s3.to_out.weight.mul_(2.0e5 / float(s3.to_out.weight.abs().max()))
            xa.to_q.weight.mul_(1.0e-6 / float(xa.to_q.weight.abs().max()))      # (to_q and to_out of a block are judged together)
        guarded, plain = run('oq'), run(False)
        with torch.no_grad():
            s3.to_out.weight.copy_(saved[0]); xa.to_q.weight.copy_(saved[1])
        assert bool(torch.isfinite(guarded).all()) and torch.equal(guarded, plain), 'out-of-range to_out / to_q weights must keep the hi + lo products'
    finally:
        K.set_proj_f16x2(os.environ.get('AMDNUWA_F16X2', K.DEFAULT_F16X2))
        A.set_precision('bf16')


def test_fp16_gradient_backward_of_the_compliant_mode(A, O):
    """round 5: blocks whose class is switched on (kernels.set_bwd_f16) keep ONE fp16 copy of their activations and run their backward on
    fp16 gradients fp16(S g), S taken on the device from the gradient that enters the pass.  One cfg-3 decoder layer at n = 2560:
      * the forward is bit-identical with the switch on and off (same products, fewer copies);
      * every gradient stays inside the mode's gradient bound against the oracle, and is no worse than the bf16 backward's;
      * the result does not depend on the magnitude of the incoming gradient (1e-6 ... 1e3 x): the scale is dynamic;
      * nothing saturates."""
    import nuwa_pytorch_amd.nuwa_pytorch as M
    from nuwa_pytorch_amd import kernels as KK
    dim, video_shape, kernel, dil, T, b = 512, (10, 16, 16), (5, 3, 3), 2, 256, 1
    torch.manual_seed(0)
    tr = M.Transformer(dim=dim, depth=1, causal=True, heads=8, dim_head=64, cross_attend=True, sparse_3dna_attn=True,
                       sparse_3dna_kernel_size=kernel, sparse_3dna_video_shape=video_shape, sparse_3dna_dilations=(dil,),
                       shift_video_tokens=True)
    with torch.no_grad():
        for n_, p in tr.named_parameters():
            if 'norm' in n_ or n_.endswith('.bias'):
                p.add_(0.1 * torch.randn_like(p))
    P = _cpu_params(tr)
    n = video_shape[0] * video_shape[1] * video_shape[2]
    g = torch.Generator().manual_seed(7)
    x = torch.randn(b, n, dim, generator=g)
    ctx = torch.randn(b, T, dim, generator=g)
    mask = torch.ones(b, T, dtype=torch.bool)
    dy = torch.randn(b, n, dim, generator=g)
    cfg = dict(video_shape=video_shape, kernel_size=kernel, dilations=(dil,), heads=8, depth=1, shift=True)
    Pr = _req(P)
    xr, cr = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    yr = O.decoder_layer(xr, O.sub(Pr, 'layers.0'), cfg, 0, cr, mask)
    yr.backward(dy)
    tr = tr.to(DEV)
    A.set_precision('bf16x3-fwd')
    saved = KK._BWD_F16
    res = {}
    try:
        def run(classes, c):
            KK.set_bwd_f16(classes)
            tr.zero_grad(set_to_none=True)
            xd, cd = x.to(DEV).requires_grad_(True), ctx.to(DEV).requires_grad_(True)
            y = tr.forward_layers(xd, context=cd, context_mask=mask.to(DEV))
            y.backward(dy.to(DEV) * c)
            worst = max(rel_err(xd.grad / c, xr.grad), rel_err(cd.grad / c, cr.grad))
            for k, p in tr.named_parameters():
                if Pr[k].grad is not None:
                    worst = max(worst, rel_err(p.grad / c, Pr[k].grad))
            ffw = rel_err(dict(tr.named_parameters())['layers.0.2.fn.fn.net.0.weight'].grad / c, Pr['layers.0.2.fn.fn.net.0.weight'].grad)
            return y.detach().clone(), worst, ffw
        KK.f16_sat_count()
        y0, w0, f0 = run('', 1.0)
        y1, w1, f1 = run('f', 1.0)
        assert torch.equal(y0, y1), 'the fp16-gradient switch must not change the forward'
        res.update(worst_bf16_bwd=w0, worst_f16_bwd=w1, ff1_weight_bf16_bwd=f0, ff1_weight_f16_bwd=f1)
        assert w1 <= 1.4e-2 and f1 <= 1.4e-2, res
        assert f1 <= f0 * 1.05, res                          # fp16 gradients carry 3 more bits than bf16 ones: the FF weight gradient must not get worse
        for c in (1e-6, 1e3):
            _, wc, fc = run('f', c)
            res[f'worst_f16_bwd_x{c:g}'] = wc
            assert wc <= 1.4e-2 and fc <= max(f1 * 1.5, 2e-3), res
        # round 6: the Sparse3DNA block too (class 's': q / k / v, the core's output and the LayerNorm output in front of the block as ONE fp16
        # copy each, amdnuwa_sparse3dna_bwd_f16): same forward bits, every gradient inside the bound and the block's own weight gradients no worse
        # ... and the cross-attention block (class 'x': the LayerNorm output, q and the core's output as ONE fp16 copy each, amdnuwa_xattn6_bwd_f16,
        # the dK / dV products on fp16 chunk-major arrays)
        for cls, wname in (('fs', 'layers.0.0.fn.fn.to_q.weight'), ('fsx', 'layers.0.1.fn.to_q.weight')):
            y2, w2, _ = run(cls, 1.0)
            assert torch.equal(y0, y2), f'the fp16-gradient switch {cls!r} must not change the forward'
            q2 = rel_err(dict(tr.named_parameters())[wname].grad, Pr[wname].grad)
            run('f', 1.0)
            q1 = rel_err(dict(tr.named_parameters())[wname].grad, Pr[wname].grad)
            res.update({f'worst_f16_bwd[{cls}]': w2, f'to_q_weight_f16_bwd[{cls}]': q2, f'to_q_weight_bf16_bwd[{cls}]': q1})
            assert w2 <= 1.4e-2 and q2 <= max(q1 * 1.05, 2e-3), res
            for c in (1e-6, 1e3):
                _, wc, _ = run(cls, c)
                res[f'worst_f16_bwd[{cls}]_x{c:g}'] = wc
                assert wc <= 1.4e-2, res
        assert KK.f16_sat_count() == 0, 'fp16 stores saturated'
    finally:
        KK._BWD_F16 = saved
        A.set_precision('bf16')
    _note('cfg3.fp16_gradient_backward', res)


def test_cfg3_full_depth_logits_worst_of_several_models_and_samples(A, O):
    """round 5: the 1e-3 logits bound of the compliant mode asserted on the WORST of eight samples, not on one sample of one random-init model:
    two random initialisations (three samples each) and a 'trained-like' variant (heavier weight tails on 1 % of the entries, LayerNorm gains
    drawn from [0.3, 3]; two samples) -- tools/parity_sweep.py.  Measured in round 5 (profiles/r05a_parity_sweep.txt): 7.2e-4 ... 8.2e-4 on the
    random initialisations, 7.5e-4 / 9.3e-4 on the trained-like model, rel-l2 6.5e-4 ... 7.6e-4.
    A second, HARSHER trained-like model (x4 tails on 2 % of the entries, LayerNorm gains in [0.1, 8]: attention scores ~64x larger, so an
    11-bit operand moves the softmax by percents) is measured and recorded, not asserted against 1e-3: no form with fp16 / bf16 operands holds
    there (5e-3 ... 7e-3 with and without the two-MFMA products) -- the mode for such weights is 'bf16x3', whose figure is recorded beside it."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import parity_sweep as PS
    from nuwa_pytorch_amd import kernels as KK
    res = PS.sweep(classes=(KK.DEFAULT_F16X2,), models=[PS.MODELS[0], PS.MODELS[1], PS.MODELS[3]], log=lambda *_: None)
    worst = max(res, key=lambda e: e['rel_max'])
    for e in res:
        record(f"cfg3.full24[bf16x3-fwd].logits[{e['model'][:24]}, sample {e['sample']}]", e['rel_max'], e['rel_l2'], 1e-3)
    harsh = PS.sweep(classes=(KK.DEFAULT_F16X2,), models=[PS.MODELS[2]], modes=('bf16x3-fwd', 'bf16x3'), log=lambda *_: None)
    _note('cfg3.full_depth_logits_sweep', dict(samples=res, worst=worst, harsh_model=harsh))
    assert all(e['finite'] for e in res + harsh)
    assert all(e['f16_saturations'] == 0 for e in res), 'fp16 stores saturated on an in-range model'
    assert len(res) == 8 and worst['rel_max'] <= 1e-3, worst
    assert max(e['rel_l2'] for e in res) <= 8.5e-4, res
    hx3 = [e for e in harsh if e['mode'] == 'bf16x3']
    assert hx3 and max(e['rel_max'] for e in hx3) <= max(e['rel_max'] for e in harsh if e['mode'] == 'bf16x3-fwd'), harsh
