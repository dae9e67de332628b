"""Host-side mirror of the reference's nuwa_pytorch/vqgan_vae.py (vq.py): VQGanVAE with the same
constructor kwargs, methods and state-dict keys.  On the NUWA training path the VAE is a FROZEN
tokenizer (copy_for_eval + get_video_indices under no_grad, vq.py:408-458) and runs through libamdnuwa;
`forward(img, return_loss=True)` (BASELINE cfg 1, CPU plumbing) runs on torch ops, as in the reference.

`VectorQuantize` restates the third-party vector_quantize_pytorch module the reference imports
(vq.py:6, 368-378; source not vendored, not installed): PARITY UNPINNED at that boundary
(SURVEY.md section 8c).  Its buffers live under `vq._codebook.*` like the upstream module's, so the autoencoder part of a
reference checkpoint loads strictly (a checkpoint trained with the reference's default use_vgg_and_gan=True also holds
`discr.*` / `vgg.*` entries: VQGanVAE drops those on load); the flat names of this repository's earlier fixtures are accepted too.

GAN / VGG training of the VAE (vq.py:145-176, 514-543) is out of scope: `use_vgg_and_gan=True` (the
reference's default, which downloads a pretrained VGG16) builds the same autoencoder WITHOUT the
perceptual / adversarial branch and says so once.
"""
import copy
import math
import os
import warnings

import torch
import torch.nn.functional as F
from torch import nn

MList = nn.ModuleList


def exists(val):
    return val is not None


def default(val, d):
    return val if exists(val) else d


def eval_decorator(fn):
    def inner(model, *args, **kwargs):
        was_training = model.training
        model.eval()
        out = fn(model, *args, **kwargs)
        model.train(was_training)
        return out
    return inner


def split_by_prefix(prefix, kwargs):
    """(kwargs whose name starts with `prefix`, prefix stripped; the rest) -- the `vq_*` pass-through of vq.py:318"""
    picked = {k[len(prefix):]: v for k, v in kwargs.items() if k.startswith(prefix)}
    rest = {k: v for k, v in kwargs.items() if not k.startswith(prefix)}
    return picked, rest


def l2norm(t):
    return F.normalize(t, dim=-1)


def leaky_relu(p=0.1):
    return nn.LeakyReLU(0.1)           # slope is always 0.1 in the reference (vq.py:94-95)


def stable_softmax(t, dim=-1, alpha=32 ** 2):
    """vq.py:97-100: softmax of t with the row maximum removed at 1/alpha scale first (same value, tamer exponent range)"""
    scaled = t / alpha
    return ((scaled - scaled.amax(dim=dim, keepdim=True).detach()) * alpha).softmax(dim=dim)


# ---------------------------------------------------------------------------------------------------
# vector quantiser (restatement of vector_quantize_pytorch; PARITY UNPINNED)
# ---------------------------------------------------------------------------------------------------

class _Codebook(nn.Module):
    """buffer container with the upstream module's names: initted, cluster_size, embed"""

    def __init__(self, codebook_size, dim, kmeans_init, cosine):
        super().__init__()
        embed = torch.randn(codebook_size, dim)
        self.register_buffer('initted', torch.tensor([not kmeans_init]))
        self.register_buffer('cluster_size', torch.zeros(codebook_size))
        self.register_buffer('embed', l2norm(embed) if cosine else embed)


class VectorQuantize(nn.Module):
    """cosine-similarity vector quantiser with EMA codebook (restatement; PARITY UNPINNED).
    eval path: x = project_in(b (h w) c); idx = argmax_c(l2norm(x) . l2norm(embed)^T) (lowest index on
    ties); quantized = project_out(embed[idx]); loss = 0."""

    def __init__(self, dim, codebook_size, codebook_dim=None, decay=0.8, commitment_weight=1.,
                 accept_image_fmap=False, kmeans_init=False, use_cosine_sim=False, eps=1e-5, **kwargs):
        super().__init__()
        codebook_dim = default(codebook_dim, dim)
        self.project_in = nn.Linear(dim, codebook_dim) if codebook_dim != dim else nn.Identity()
        self.project_out = nn.Linear(codebook_dim, dim) if codebook_dim != dim else nn.Identity()
        self.decay, self.commitment_weight, self.eps = decay, commitment_weight, eps
        self.accept_image_fmap = accept_image_fmap
        self.use_cosine_sim = use_cosine_sim
        self._codebook = _Codebook(codebook_size, codebook_dim, kmeans_init, use_cosine_sim)

    # the three buffers, under the names the rest of the package uses
    embed = property(lambda self: self._codebook.embed)
    cluster_size = property(lambda self: self._codebook.cluster_size)
    initted = property(lambda self: self._codebook.initted)
    codebook = property(lambda self: self._codebook.embed)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """accept (i) the flat buffer names of this repository's first fixtures (`vq.embed` ...), (ii) upstream releases that
        carry a leading num_codebooks axis of 1 and an `embed_avg` EMA numerator, (iii) a float `initted` flag"""
        cb = prefix + '_codebook.'
        for name in ('embed', 'cluster_size', 'initted'):
            if prefix + name in state_dict and cb + name not in state_dict:
                state_dict[cb + name] = state_dict.pop(prefix + name)
        state_dict.pop(cb + 'embed_avg', None)
        for name, nd in (('embed', 2), ('cluster_size', 1)):
            t = state_dict.get(cb + name)
            if t is not None and t.dim() == nd + 1 and t.shape[0] == 1:
                state_dict[cb + name] = t[0]
        t = state_dict.get(cb + 'initted')
        if t is not None and t.dtype != torch.bool:
            state_dict[cb + 'initted'] = t.reshape(-1)[:1] != 0
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _nearest(self, flat):
        if self.use_cosine_sim:
            feats = l2norm(flat)
            if self.training and not bool(self.initted):          # first training batch seeds the codebook
                pick = torch.randint(0, feats.shape[0], (self.embed.shape[0],), device=feats.device)
                self.embed.copy_(feats[pick].detach())
                self.initted.fill_(True)
            return feats, (feats @ l2norm(self.embed).t()).argmax(dim=-1)
        return flat, (-torch.cdist(flat, self.embed)).argmax(dim=-1)

    @torch.no_grad()
    def _ema_update(self, feats, ind):
        hits = F.one_hot(ind, self.embed.shape[0]).type(feats.dtype)
        count = hits.sum(0)
        self.cluster_size.mul_(self.decay).add_(count, alpha=1 - self.decay)
        sums = hits.t() @ feats
        fresh = l2norm(sums) if self.use_cosine_sim else sums / count.clamp(min=1)[:, None]
        target = torch.where((count > 0)[:, None], fresh, self.embed)
        blended = self.embed * self.decay + target * (1 - self.decay)
        self.embed.copy_(l2norm(blended) if self.use_cosine_sim else blended)

    def forward(self, x):
        fmap_shape = None
        if self.accept_image_fmap:
            fmap_shape = x.shape
            x = x.flatten(2).transpose(1, 2)                      # b c h w -> b (h w) c
        x = self.project_in(x)
        feats, ind = self._nearest(x.reshape(-1, x.shape[-1]))
        quant = self.embed[ind].reshape(x.shape)
        loss = x.new_zeros(1)
        if self.training:
            self._ema_update(feats.detach(), ind)
            loss = F.mse_loss(quant.detach(), x) * self.commitment_weight
            quant = x + (quant - x).detach()                      # straight-through
        quant = self.project_out(quant)
        ind = ind.reshape(x.shape[:-1])
        if fmap_shape is not None:
            B, _, Hh, Ww = fmap_shape
            quant = quant.transpose(1, 2).reshape(B, -1, Hh, Ww)
            ind = ind.reshape(B, Hh, Ww)
        return quant, ind, loss.reshape(1)


# ---------------------------------------------------------------------------------------------------
# blocks
# ---------------------------------------------------------------------------------------------------

class LayerNormChan(nn.Module):
    """vq.py:129-143: LayerNorm over the channel axis of an NCHW map"""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))
        self.b = nn.Parameter(torch.zeros(1, dim, 1, 1))

    def forward(self, x):
        mu = x.mean(dim=1, keepdim=True)
        var = (x - mu).pow(2).mean(dim=1, keepdim=True)
        return (x - mu) * torch.rsqrt(var + self.eps) * self.g + self.b


class ContinuousPositionBias(nn.Module):
    """vq.py:178-210: an MLP maps the signed-log relative (dy, dx) of every pair of feature-map positions to one bias per head"""

    def __init__(self, *, dim, heads, layers=2):
        super().__init__()
        widths = [2] + [dim] * layers
        self.net = MList([nn.Sequential(nn.Linear(a, b), leaky_relu()) for a, b in zip(widths[:-1], widths[1:])])
        self.net.append(nn.Linear(dim, heads))
        self.register_buffer('rel_pos', None, persistent=False)

    def _coords(self, side, device):
        if self.rel_pos is None or self.rel_pos.shape[0] != side * side or self.rel_pos.device != device:
            ax = torch.arange(side, device=device)
            yx = torch.cartesian_prod(ax, ax)                                  # (side^2, 2) as (row, col)
            delta = (yx[:, None] - yx[None]).float()
            self.rel_pos = delta.sign() * (delta.abs() + 1).log()
        return self.rel_pos

    def forward(self, x):
        h = self._coords(math.isqrt(x.shape[-1]), x.device)
        for layer in self.net:
            h = layer(h)
        return x + h.movedim(-1, 0)                                           # (i, j, heads) -> (heads, i, j)


def _res_body(chan, groups, glu):
    """conv3x3 -> [GLU | GroupNorm + LeakyReLU] twice, then conv1x1: GLUResBlock keeps (conv, GLU, GroupNorm) order, ResBlock
    (conv, GroupNorm, LeakyReLU) (vq.py:212-242) -- the indices of the parameterised layers are part of the state-dict keys"""
    layers = []
    for _ in range(2):
        if glu:
            layers += [nn.Conv2d(chan, chan * 2, 3, padding=1), nn.GLU(dim=1), nn.GroupNorm(groups, chan)]
        else:
            layers += [nn.Conv2d(chan, chan, 3, padding=1), nn.GroupNorm(groups, chan), leaky_relu()]
    return nn.Sequential(*layers, nn.Conv2d(chan, chan, 1))


class GLUResBlock(nn.Module):
    def __init__(self, chan, groups=16):
        super().__init__()
        self.net = _res_body(chan, groups, glu=True)

    def forward(self, x):
        return self.net(x) + x


class ResBlock(nn.Module):
    def __init__(self, chan, groups=16):
        super().__init__()
        self.net = _res_body(chan, groups, glu=False)

    def forward(self, x):
        return self.net(x) + x


class VQGanAttention(nn.Module):
    """vq.py:244-286"""

    def __init__(self, *, dim, dim_head=64, heads=8, dropout=0.):
        super().__init__()
        self.heads = heads
        self.scale = nn.Parameter(torch.ones(1, heads, 1, 1) * math.log(0.01))
        inner_dim = heads * dim_head
        self.dropout = nn.Dropout(dropout)
        self.post_norm = LayerNormChan(dim)
        self.cpb = ContinuousPositionBias(dim=dim // 4, heads=heads)
        self.to_qkv = nn.Conv2d(dim, inner_dim * 3, 1, bias=False)
        self.to_out = nn.Conv2d(inner_dim, dim, 1)

    def forward(self, x):
        B, _, height, width = x.shape
        q, k, v = (t.reshape(B, self.heads, -1, height * width) for t in self.to_qkv(x).chunk(3, dim=1))
        q, k = l2norm(q), l2norm(k)                          # over the SPATIAL axis (quirk Q9)
        sim = torch.matmul(q.transpose(-1, -2), k) * self.scale.exp()
        attn = self.dropout(stable_softmax(self.cpb(sim), dim=-1))
        out = torch.matmul(v, attn.transpose(-1, -2)).reshape(B, -1, height, width)
        return self.post_norm(self.to_out(out)) + x


# ---------------------------------------------------------------------------------------------------
# the autoencoder
# ---------------------------------------------------------------------------------------------------

class VQGanVAE(nn.Module):
    """vq.py:288-548 (GAN/VGG branch excluded)."""


    def __init__(self, *, dim, image_size, channels=3, num_layers=4, layer_mults=None, l2_recon_loss=False,
                 use_hinge_loss=True, num_resnet_blocks=1, vgg=None, vq_codebook_dim=256, vq_codebook_size=512,
                 vq_decay=0.8, vq_commitment_weight=1., vq_kmeans_init=True, vq_use_cosine_sim=True, use_attn=True,
                 attn_dim_head=64, attn_heads=8, resnet_groups=16, attn_dropout=0., first_conv_kernel_size=5,
                 use_vgg_and_gan=True, **kwargs):
        super().__init__()
        assert dim % resnet_groups == 0, f'dimension {dim} must be divisible by {resnet_groups} (groups for the groupnorm)'
        if use_vgg_and_gan:          # (per instance: every VAE built this way optimises a different objective from the reference's default)
            warnings.warn('nuwa_pytorch_amd.VQGanVAE: the perceptual (VGG16) and adversarial (discriminator) losses of the '
                          'reference are outside this package; building the autoencoder with use_vgg_and_gan=False '
                          '(forward(return_loss=True) = reconstruction loss only).  NUWA uses the VAE frozen, so training it is '
                          'unaffected; `discr.*` / `vgg.*` entries of a reference checkpoint are skipped on load.', stacklevel=2)
            use_vgg_and_gan = False
        vq_kwargs, kwargs = split_by_prefix('vq_', kwargs)
        self.image_size = image_size
        self.channels = channels
        self.num_layers = num_layers
        self.fmap_size = image_size // (num_layers ** 2)        # as in the reference (quirk Q6)
        self.codebook_size = vq_codebook_size
        layer_mults = default(layer_mults, [2 ** i for i in range(num_layers)])
        assert len(layer_mults) == num_layers, 'layer multipliers must be equal to designated number of layers'
        widths = [dim] + [dim * m for m in layer_mults]
        # a scalar setting applies to the deepest level only (vq.py:339-345)
        res_counts = num_resnet_blocks if isinstance(num_resnet_blocks, tuple) else (0,) * (num_layers - 1) + (num_resnet_blocks,)
        attn_flags = use_attn if isinstance(use_attn, tuple) else (False,) * (num_layers - 1) + (use_attn,)
        assert len(res_counts) == num_layers and len(attn_flags) == num_layers
        attn = lambda w: VQGanAttention(dim=w, heads=attn_heads, dim_head=attn_dim_head, dropout=attn_dropout)
        # stage lists in execution order.  Encoder level i: stride-2 conv, its ResBlocks, its attention.  The decoder mirrors it
        # (attention / GLU ResBlocks first, then x2 upsample + conv), deepest level first.  Sub-modules of one level are created
        # in the reference's order (vq.py:349-363), so a seeded default initialisation draws the same numbers.
        enc, dec = [], []
        for w_in, w_out, n_res, with_attn in zip(widths[:-1], widths[1:], res_counts, attn_flags):
            enc.append(nn.Sequential(nn.Conv2d(w_in, w_out, 4, stride=2, padding=1), leaky_relu()))
            level = [nn.Sequential(nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False),
                                   nn.Conv2d(w_out, w_in, 3, padding=1), leaky_relu())]
            if with_attn:
                level.insert(0, attn(w_out))
            for _ in range(n_res):
                enc.append(ResBlock(w_out, groups=resnet_groups))
                level.insert(0, GLUResBlock(w_out, groups=resnet_groups))
            if with_attn:
                enc.append(attn(w_out))
            dec = level + dec
        stem = nn.Conv2d(channels, dim, first_conv_kernel_size, padding=first_conv_kernel_size // 2)
        self.encoders = MList([stem] + enc)
        self.decoders = MList(dec + [nn.Conv2d(dim, channels, 1)])
        self.vq = VectorQuantize(dim=widths[-1], codebook_dim=vq_codebook_dim, codebook_size=vq_codebook_size,
                                 decay=vq_decay, commitment_weight=vq_commitment_weight, accept_image_fmap=True,
                                 kmeans_init=vq_kmeans_init, use_cosine_sim=vq_use_cosine_sim, **vq_kwargs)
        self.recon_loss_fn = F.mse_loss if l2_recon_loss else F.l1_loss
        self.vgg = None
        self.discr = None
        self.use_vgg_and_gan = use_vgg_and_gan

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        """a reference checkpoint written with use_vgg_and_gan=True also carries the discriminator and the VGG16 (`discr.*`,
        `vgg.*`, vq.py:396-406): this package builds neither, so those entries are dropped before the (strict) load"""
        kept = {k: v for k, v in state_dict.items() if not (k.startswith('discr.') or k.startswith('vgg.'))}
        return super().load_state_dict(kept, strict=strict, **kwargs)

    def copy_for_eval(self):
        device = next(self.parameters()).device
        vae_copy = copy.deepcopy(self.cpu())
        vae_copy.eval()
        return vae_copy.to(device)

    @property
    def codebook(self):
        return self.vq.codebook

    def encode(self, fmap):
        for enc in self.encoders:
            fmap = enc(fmap)
        return self.vq(fmap)

    def codes_for_decoder(self, indices):
        """codebook rows for `indices`, in the channel width the decoder takes.  The reference indexes the raw codebook
        (vq.py:447, np.py:1910), which only fits the decoder when vq_codebook_dim equals the last encoder width; otherwise
        (e.g. the default codebook_dim 256 under a 512-wide cfg-3 encoder) its decode raises.  Here the quantiser's own
        project_out -- what VQGanVAE.forward feeds the decoder -- bridges the two; it is the identity when the widths agree."""
        return self.vq.project_out(self.codebook[indices])

    def decode(self, fmap):
        for dec in self.decoders:
            fmap = dec(fmap)
        return fmap

    def _hip_decode(self, fmap):
        fmap = fmap.float()
        for dec in self.decoders:
            fmap = self._hip_module(dec, fmap)
        return fmap

    @torch.no_grad()
    @eval_decorator
    def codebook_indices_to_video(self, indices):
        b = indices.shape[0]
        codes = self.codes_for_decoder(indices)
        fs = self.fmap_size
        codes = codes.reshape(b, -1, fs, fs, codes.shape[-1]).permute(0, 1, 4, 2, 3).reshape(-1, codes.shape[-1], fs, fs)
        video = self._hip_decode(codes) if codes.is_cuda else self.decode(codes)     # sampling path (np.py:1912): libamdnuwa on device
        return video.reshape(b, -1, *video.shape[1:])

    def _hip_module(self, m, x):
        """one encoder stage through libamdnuwa (exact fp32): convolutions, GroupNorm(+LeakyReLU), and the VQGanAttention
        block (q/k l2norm over the spatial axis, biased softmax attention, LayerNormChan + residual)."""
        from . import kernels as K
        if isinstance(m, nn.Conv2d):
            assert m.stride[0] == m.stride[1] and m.padding[0] == m.padding[1] and m.dilation == (1, 1) and m.groups == 1
            return K.conv2d_fwd(x, m.weight, m.bias, m.stride[0], m.padding[0])
        if isinstance(m, nn.Sequential) and len(m) == 2 and isinstance(m[0], nn.Conv2d) and isinstance(m[1], nn.LeakyReLU):
            c = m[0]
            return K.conv2d_fwd(x, c.weight, c.bias, c.stride[0], c.padding[0], leaky=True)
        if isinstance(m, ResBlock):
            c1, g1, _, c2, g2, _, c3 = m.net
            h = K.conv2d_fwd(x, c1.weight, c1.bias, 1, 1)
            h = K.groupnorm_fwd(h, g1.weight, g1.bias, g1.num_groups, g1.eps, leaky=True)
            h = K.conv2d_fwd(h, c2.weight, c2.bias, 1, 1)
            h = K.groupnorm_fwd(h, g2.weight, g2.bias, g2.num_groups, g2.eps, leaky=True)
            return K.conv2d_fwd(h, c3.weight, c3.bias, 1, 0) + x
        if isinstance(m, VQGanAttention):
            B, _, height, width = x.shape
            P_ = height * width
            # continuous position bias: a function of the module's parameters only (vq.py:192-226) -> [heads, P, P]
            bias = m.cpb(torch.zeros(1, m.heads, P_, P_, device=x.device))[0]
            return K.vqgan_attention(x, m.to_qkv.weight, m.to_out.weight, m.to_out.bias, bias, m.scale, m.post_norm.g, m.post_norm.b,
                                     m.heads, m.post_norm.eps)
        if isinstance(m, GLUResBlock):                     # decoder (vq.py:212-226)
            c1, _, g1, c2, _, g2, c3 = m.net
            h = K.groupnorm_fwd(K.glu_chan(K.conv2d_fwd(x, c1.weight, c1.bias, 1, 1)), g1.weight, g1.bias, g1.num_groups, g1.eps)
            h = K.groupnorm_fwd(K.glu_chan(K.conv2d_fwd(h, c2.weight, c2.bias, 1, 1)), g2.weight, g2.bias, g2.num_groups, g2.eps)
            return K.conv2d_fwd(h, c3.weight, c3.bias, 1, 0) + x
        if isinstance(m, nn.Sequential) and len(m) == 3 and isinstance(m[0], nn.Upsample) and isinstance(m[1], nn.Conv2d):
            c = m[1]                                       # decoder stage: x2 bilinear upsample, conv 3x3, LeakyReLU
            return K.conv2d_fwd(K.upsample_bilinear2x(x), c.weight, c.bias, c.stride[0], c.padding[0], leaky=True)
        raise NotImplementedError(f'no libamdnuwa path for VAE stage {type(m).__name__}')

    def _hip_encode_indices(self, images):
        from . import kernels as K
        if not self.vq.use_cosine_sim:
            raise NotImplementedError('libamdnuwa VQ lookup implements the cosine-similarity codebook (vq_use_cosine_sim=True)')
        fmap = images.float()
        for enc in self.encoders:
            fmap = self._hip_module(enc, fmap)
        B, _, Hh, Ww = fmap.shape
        pin = self.vq.project_in
        if isinstance(pin, nn.Linear):
            fmap = K.conv2d_fwd(fmap, pin.weight[:, :, None, None], pin.bias, 1, 0)
        rows = fmap.permute(0, 2, 3, 1).reshape(B * Hh * Ww, -1)
        return K.vq_argmax(rows, self.vq.embed).reshape(B, Hh, Ww)

    @torch.no_grad()
    @eval_decorator
    def get_video_indices(self, video):
        """frozen tokenizer of the NUWA training step (reference vqgan_vae.py:452-458): runs on the MI355X through
        libamdnuwa; there is no CPU path."""
        if not video.is_cuda:
            raise RuntimeError('VQGanVAE.get_video_indices runs through libamdnuwa and needs the video on an MI355X device')
        b, f, _, h, w = video.shape
        images = video.reshape(b * f, *video.shape[2:])
        # every stage works per image (conv, GroupNorm, attention, VQ lookup): chunks of the frame list give the same ids bit for bit and keep the
        # tokenizer's fp32 feature maps (5 GiB per 1280 frames after the first conv) out of a step that already fills the card (tools/full_step.py,
        # b = 128: 235 GiB).  AMDNUWA_TOKENIZER_CHUNK = frames per call (0 = all at once)
        chunk = int(os.environ.get('AMDNUWA_TOKENIZER_CHUNK', '320'))
        if chunk <= 0 or images.shape[0] <= chunk:
            indices = self._hip_encode_indices(images)
        else:
            indices = torch.cat([self._hip_encode_indices(images[i:i + chunk]) for i in range(0, images.shape[0], chunk)], dim=0)
        return indices.reshape(b, f, *indices.shape[1:])

    def forward(self, img, return_loss=False, return_discr_loss=False, return_recons=False, apply_grad_penalty=False):
        batch, channels, height, width = img.shape
        assert height == self.image_size and width == self.image_size, 'height and width of input image must be equal to {self.image_size}'
        assert channels == self.channels, 'number of channels on image or sketch is not equal to the channels set on this VQGanVAE'
        fmap, indices, commit_loss = self.encode(img)
        fmap = self.decode(fmap)
        if not return_loss and not return_discr_loss:
            return fmap
        assert return_loss ^ return_discr_loss, 'you should either return autoencoder loss or discriminator loss, but not both'
        assert not return_discr_loss, 'discriminator must exist to train it (the adversarial branch is outside this package)'
        recon_loss = self.recon_loss_fn(fmap, img)        # only term when use_vgg_and_gan=False (quirk Q17)
        if return_recons:
            return recon_loss, fmap
        return recon_loss
