"""NUWASketch pieces that run on PyTorch ops (row f4): SparseCross2DNA and the non-causal Sparse3DNA of the sketch encoder,
checked against the reference's own modules (imported read-only through oracle/ref_shims.py; build container only -- on the
GPU box the committed fixture tests/golden/g11_sketch.npz, generated from the same import, takes over)."""
import pytest
import torch

pytestmark = pytest.mark.reference

TOL = dict(rtol=1e-4, atol=2e-5)


def _grads(mod, out, dy, xs):
    for p in mod.parameters():
        p.grad = None
    gx = torch.autograd.grad(out, xs, dy, retain_graph=True)
    out.backward(dy)
    return gx, {k: (p.grad.clone() if p.grad is not None else None) for k, p in mod.named_parameters()}


@pytest.mark.parametrize('fmap,frames,kernel,dil,n,masked', [(4, 2, 3, 1, None, False), (4, 2, 3, 2, None, True), (4, 1, 5, 1, 20, True),
                                                           (8, 3, 3, 1, 70, True), (4, 2, 3, 1, 1, False), (4, 2, 3, 1, 2, True)])
def test_sparse_cross_2dna_matches_reference(reference_pkg, fmap, frames, kernel, dil, n, masked):
    from nuwa_pytorch.nuwa_pytorch import SparseCross2DNA as Ref
    from nuwa_pytorch_amd.nuwa_pytorch import SparseCross2DNA as Mine
    torch.manual_seed(0)
    kw = dict(dim=32, image_size=fmap, heads=2, dim_head=16, kernel_size=kernel, dilation=dil)
    ref, mine = Ref(**kw), Mine(**kw)
    assert set(mine.state_dict()) == set(ref.state_dict())
    mine.load_state_dict(ref.state_dict())
    n = 1 + 3 * fmap * fmap if n is None else n
    x = torch.randn(2, n, 32, requires_grad=True)
    ctx = torch.randn(2, frames * fmap * fmap, 32, requires_grad=True)
    cmask = None
    if masked:
        cmask = torch.rand(2, frames * fmap * fmap) > 0.3
        cmask[0] = False                                   # condition-dropped sample
    yr = ref(x, context=ctx, context_mask=cmask)
    ym = mine(x, context=ctx, context_mask=cmask)
    torch.testing.assert_close(ym, yr, **TOL)
    dy = torch.randn_like(yr)
    (gxr, gcr), gr = _grads(ref, yr, dy, (x, ctx))
    (gxm, gcm), gm = _grads(mine, ym, dy, (x, ctx))
    torch.testing.assert_close(gxm, gxr, **TOL)
    torch.testing.assert_close(gcm, gcr, **TOL)
    for k in gr:
        assert (gm[k] is None) == (gr[k] is None), k
        if gr[k] is not None:
            torch.testing.assert_close(gm[k], gr[k], **TOL)


@pytest.mark.parametrize('shape,kernel,dil,n,rel', [((2, 4, 4), 3, 1, None, False), ((3, 4, 4), (3, 3, 3), (1, 2, 1), None, False),
                                                    ((2, 4, 4), 3, 1, 20, False), ((2, 4, 4), 3, 1, 1, False), ((2, 4, 4), 3, 1, 17, False),
                                                    ((2, 4, 4), 3, 2, None, False)])
def test_noncausal_sparse3dna_matches_reference(reference_pkg, shape, kernel, dil, n, rel):
    from nuwa_pytorch.nuwa_pytorch import Sparse3DNA as Ref
    from nuwa_pytorch_amd.nuwa_pytorch import Sparse3DNA as Mine
    torch.manual_seed(1)
    kw = dict(dim=32, video_shape=shape, kernel_size=kernel, dilation=dil, heads=2, dim_head=16, causal=False, rel_pos_bias=rel)
    ref, mine = Ref(**kw), Mine(**kw)
    assert torch.equal(mine.mask, ref.mask)
    mine.load_state_dict(ref.state_dict())
    N = shape[0] * shape[1] * shape[2]
    n = N if n is None else n                              # the sketch encoder feeds exactly f*h*w tokens (no <bos>)
    x = torch.randn(2, n, 32, requires_grad=True)
    yr, ym = ref(x), mine(x)
    torch.testing.assert_close(ym, yr, **TOL)
    dy = torch.randn_like(yr)
    (gxr,), gr = _grads(ref, yr, dy, (x,))
    (gxm,), gm = _grads(mine, ym, dy, (x,))
    torch.testing.assert_close(gxm, gxr, **TOL)
    for k in gr:
        assert (gm[k] is None) == (gr[k] is None), k
        if gr[k] is not None:
            torch.testing.assert_close(gm[k], gr[k], **TOL)
