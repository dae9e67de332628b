"""Frozen VQGanVAE tokenizer on the MI355X: the exact-fp32 HIP kernels behind
VQGanVAE.get_video_indices against torch fp32 on the CPU, the oracle, and the golden fixture g7_vae captured
from the reference.  Feature maps: 1e-5 relative (fp32, different summation order).  VQ indices: bit-exact
wherever the top-2 similarity gap exceeds 1e-5 (in practice: everywhere)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from golden_util import load  # noqa: E402
from gpu_util import report  # noqa: E402

DEV = 'cuda'


@pytest.fixture(scope='module')
def K():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nuwa_pytorch_amd import kernels
    return kernels


@pytest.mark.parametrize('N,Cin,H,W,Cout,k,stride,pad,leaky', [
    (2, 3, 32, 32, 32, 5, 1, 2, False),        # first conv (vqgan_vae.py:365)
    (2, 32, 32, 32, 64, 4, 2, 1, True),        # strided encoder conv + LeakyReLU (vqgan_vae.py:352)
    (3, 64, 8, 8, 64, 3, 1, 1, False),         # ResBlock 3x3
    (3, 64, 8, 8, 192, 1, 1, 0, False),        # 1x1 (to_qkv / project_in)
    (1, 5, 7, 9, 130, 3, 1, 1, True),          # ragged: odd sizes, Cout > one tile
    (1, 128, 16, 16, 256, 4, 2, 1, True),      # cfg-3 sized channel pair
    (2, 6, 9, 11, 70, 2, 1, 0, False),         # kernel size without a compiled specialisation (generic k-split), Cout > 64
    (1, 4, 12, 12, 8, 7, 1, 3, True),          # 7x7 (generic), narrow tile, K = 196 (a multiple of 4: vector weight reads)
    (1, 3, 10, 10, 20, 3, 2, 1, False),        # K = 27: scalar weight reads, strided
])
def test_conv2d(K, N, Cin, H, W, Cout, k, stride, pad, leaky):
    torch.manual_seed(0)
    x, w, b = torch.randn(N, Cin, H, W), torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5, torch.randn(Cout)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    if leaky:
        ref = F.leaky_relu(ref, 0.1)
    y = K.conv2d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), stride, pad, leaky=leaky)
    report(f'conv2d[{Cin}->{Cout},k{k},s{stride}]', y, ref, 1e-5)
    y2 = K.conv2d_fwd(x.to(DEV), w.to(DEV), None, stride, pad, leaky=False)
    report(f'conv2d_nobias[{Cin}->{Cout},k{k}]', y2, F.conv2d(x, w, None, stride=stride, padding=pad), 1e-5)


@pytest.mark.parametrize('N,C,H,G,leaky', [(2, 64, 8, 16, True), (3, 32, 5, 16, False), (1, 512, 16, 16, True)])
def test_groupnorm(K, N, C, H, G, leaky):
    torch.manual_seed(1)
    x, w, b = torch.randn(N, C, H, H) * 2 + 0.5, torch.randn(C), torch.randn(C)
    ref = F.group_norm(x, G, w, b, 1e-5)
    if leaky:
        ref = F.leaky_relu(ref, 0.1)
    report(f'groupnorm[{C}/{G}]', K.groupnorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), G, 1e-5, leaky=leaky), ref, 1e-5)


@pytest.mark.parametrize('R,Cn,Dc', [(100, 64, 16), (2560, 8192, 256), (1, 5, 2), (64, 130, 32), (131, 1000, 256), (20480, 8192, 256), (7, 33, 256)])
def test_vq_argmax(K, O_, R, Cn, Dc):
    torch.manual_seed(2)
    x, cb = torch.randn(R, Dc), torch.randn(Cn, Dc)
    idx_ref, gap = (t.reshape(-1) for t in O_.vq_eval_lookup(x.t()[None, :, :, None], cb))
    idx, sim = K.vq_argmax(x.to(DEV), cb.to(DEV), want_sim=True)
    sure = gap > 1e-5
    assert torch.equal(idx.cpu()[sure], idx_ref[sure])
    assert float(sure.float().mean()) > 0.99
    ref_sim = (F.normalize(x, dim=-1) * F.normalize(cb, dim=-1)[idx_ref]).sum(-1)
    report(f'vq_sim[{R}x{Cn}x{Dc}]', sim, ref_sim, 1e-5)


def test_vq_argmax_lowest_index_on_exact_ties(K):
    """duplicate codes give bit-identical similarities: the lowest index must win, as torch.argmax does"""
    torch.manual_seed(3)
    cb = torch.randn(70, 32)
    cb[65] = cb[3]
    cb[40] = cb[3]
    cb[69] = cb[17]
    x = torch.cat([cb[3:4] * 2.5, cb[17:18] * 0.3, torch.randn(5, 32)])
    idx = K.vq_argmax(x.to(DEV), cb.to(DEV)).cpu()
    assert idx[0] == 3 and idx[1] == 17


def test_vq_argmax_sliced_form_ties_and_agreement(K):
    """code_dim 256 runs the register-resident-rows kernel with the code axis cut in slices: duplicates placed in different slices and
    tiles must still resolve to the lowest index, and both kernel forms must pick the same codes with the same similarity"""
    from nuwa_pytorch_amd import _lib
    torch.manual_seed(4)
    cb = torch.randn(8192, 256)
    for dup in (31, 32, 700, 4097, 8191):
        cb[dup] = cb[5]
    cb[8000] = cb[6000]
    x = torch.cat([cb[5:6] * 3.0, cb[6000:6001] * 0.2, torch.randn(300, 256)]).to(DEV)
    idx, sim = K.vq_argmax(x, cb.to(DEV), want_sim=True)
    assert int(idx[0]) == 5 and int(idx[1]) == 6000
    L = _lib.lib()
    L.amdnuwa_set_tuning(15, 1)
    try:
        idx1, sim1 = K.vq_argmax(x, cb.to(DEV), want_sim=True)
    finally:
        L.amdnuwa_set_tuning(15, 0)
    assert torch.equal(idx, idx1)
    report('vq_sim[sliced vs first form]', sim, sim1.cpu(), 1e-5)


def test_vqgan_attention_mfma_form_at_cfg3_shape(K):
    """VQGanAttention at the cfg-3 shape (512 channels, 8 heads x 64, 16 x 16 positions) runs the transposed f32-MFMA core and the
    register LayerNormChan; against the torch module on the CPU and against the first (VALU) form of the same kernels"""
    import nuwa_pytorch_amd as A
    from nuwa_pytorch_amd import _lib
    from nuwa_pytorch_amd.vqgan_vae import VQGanAttention
    torch.manual_seed(5)
    m = VQGanAttention(dim=512, dim_head=64, heads=8).eval()
    with torch.no_grad():
        m.scale.add_(torch.randn_like(m.scale) * 0.3 + 3.0)      # sharpen the softmax: a flat one would hide indexing errors
        m.post_norm.g.mul_(torch.rand_like(m.post_norm.g) + 0.5)
        x = torch.randn(3, 512, 16, 16)
        ref = m(x)
        vae = A.VQGanVAE(dim=32, image_size=32, num_layers=2, vq_codebook_size=64, vq_codebook_dim=16, use_vgg_and_gan=False)
        md = m.to(DEV)
        y = vae._hip_module(md, x.to(DEV))
        L = _lib.lib()
        L.amdnuwa_set_tuning(15, 1)
        try:
            y1 = vae._hip_module(md, x.to(DEV))
        finally:
            L.amdnuwa_set_tuning(15, 0)
    report('vqgan_attention[mfma]', y, ref, 2e-5)
    report('vqgan_attention[mfma vs valu]', y, y1.cpu(), 2e-5)


@pytest.fixture(scope='module')
def O_():
    from oracle import nuwa_oracle
    return nuwa_oracle


def test_g7_vae_tokenizer_against_reference_fixture(K):
    import nuwa_pytorch_amd as A
    Ar, P, _ = load('g7_vae')
    vae = A.VQGanVAE(dim=32, image_size=32, num_layers=2, vq_codebook_size=64, vq_codebook_dim=16, use_vgg_and_gan=False,
                     attn_dim_head=16, attn_heads=4)
    vae.load_state_dict(P)
    vae = vae.to(DEV).eval()
    fm = Ar['img'].to(DEV)
    with torch.no_grad():
        for i, enc in enumerate(vae.encoders):
            fm = vae._hip_module(enc, fm)
            report(f'g7.stage{i}', fm, Ar[f'stage{i}'], 2e-5)
    video = Ar['img'].to(DEV)[None]                       # [1, f=3, c, h, w]
    idx = vae.get_video_indices(video)[0].cpu()
    sure = Ar['top2_gap'].reshape(idx.shape) > 1e-5
    assert torch.equal(idx[sure], Ar['indices'].reshape(idx.shape)[sure])
    assert bool(sure.all()), 'fixture has near-ties'
    with pytest.raises(RuntimeError):
        vae.get_video_indices(Ar['img'][None])           # no CPU path


def test_cfg3_vae_tokenizer_matches_oracle():
    """full-size cfg-3 VAE (dim 64, 256x256 frames, 4 layers, codebook 8192 x 256): one frame against the oracle"""
    import nuwa_pytorch_amd as A
    from oracle import nuwa_oracle as O
    torch.manual_seed(0)
    vae = A.VQGanVAE(dim=64, image_size=256, num_layers=4, vq_codebook_size=8192, use_vgg_and_gan=False).eval()
    img = torch.rand(1, 3, 256, 256)
    P = {k: v.detach() for k, v in vae.state_dict().items()}
    fm = O.vae_encode_fmap(img, P, num_layers=4, heads=8)
    idx_ref, gap = (t.reshape(-1) for t in O.vq_eval_lookup(fm, P['vq._codebook.embed'], P['vq.project_in.weight'], P['vq.project_in.bias']))
    idx = vae.to(DEV).get_video_indices(img.to(DEV)[None])[0, 0].reshape(-1).cpu()
    sure = gap > 1e-5
    assert float(sure.float().mean()) > 0.98
    assert torch.equal(idx[sure], idx_ref[sure])


def test_tokenizer_in_frame_chunks_gives_the_same_ids(monkeypatch):
    """VQGanVAE.get_video_indices walks the frame list in chunks (AMDNUWA_TOKENIZER_CHUNK frames per call, default 320): every stage works per
    image, so the ids must be the ids of one call over all frames, bit for bit -- also with a ragged last chunk"""
    import nuwa_pytorch_amd as A
    torch.manual_seed(3)
    vae = A.VQGanVAE(dim=32, image_size=64, num_layers=2, vq_codebook_size=512, use_vgg_and_gan=False).eval().to(DEV)
    video = torch.rand(3, 5, 3, 64, 64, device=DEV)
    monkeypatch.setenv('AMDNUWA_TOKENIZER_CHUNK', '0')
    whole = vae.get_video_indices(video)
    for chunk in ('4', '15', '7'):
        monkeypatch.setenv('AMDNUWA_TOKENIZER_CHUNK', chunk)
        assert torch.equal(vae.get_video_indices(video), whole), chunk


def test_g7_vae_decoder_on_hip(K):
    """VQGanVAE.decode (vq.py:437-441) through libamdnuwa: GLUResBlock, VQGanAttention, x2 bilinear upsample + conv stages,
    final 1x1 conv, against the reference's reconstruction in fixture g7 (decode of the quantised feature map)"""
    import nuwa_pytorch_amd as A
    Ar, P, _ = load('g7_vae')
    vae = A.VQGanVAE(dim=32, image_size=32, num_layers=2, vq_codebook_size=64, vq_codebook_dim=16, use_vgg_and_gan=False,
                     attn_dim_head=16, attn_heads=4)
    vae.load_state_dict(P)
    vae = vae.to(DEV).eval()
    with torch.no_grad():
        ind = Ar['indices'].to(DEV)                                      # [3, 8, 8]
        quant = vae.vq.project_out(vae.vq.embed[ind]).permute(0, 3, 1, 2).contiguous()
        rec = vae._hip_decode(quant)
    report('g7.recon', rec, Ar['recon'], 5e-5)
    x = torch.randn(2, 6, 5, 7, device=DEV)
    report('upsample2x', K.upsample_bilinear2x(x), F.interpolate(x.cpu(), scale_factor=2, mode='bilinear', align_corners=False), 1e-6)
    report('glu', K.glu_chan(x), F.glu(x.cpu(), dim=1), 1e-6)
