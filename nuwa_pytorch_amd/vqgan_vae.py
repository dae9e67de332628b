"""Host-side mirror of the reference's nuwa_pytorch/vqgan_vae.py (vq.py): VQGanVAE with the same
constructor kwargs, methods and state-dict keys.  On the NUWA training path the VAE is a FROZEN
tokenizer (copy_for_eval + get_video_indices under no_grad, vq.py:408-458).

`VectorQuantize` restates the third-party vector_quantize_pytorch module the reference imports
(vq.py:6, 368-378; source not vendored, not installed): PARITY UNPINNED at that boundary
(SURVEY.md section 8c).  GAN / VGG training of the VAE (vq.py:145-176, 514-543) is out of scope.
"""
import copy
import math
from functools import partial
from math import sqrt

import torch
import torch.nn.functional as F
from torch import nn, einsum

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


def group_dict_by_key(cond, d):
    return_val = [dict(), dict()]
    for key in d.keys():
        match = bool(cond(key))
        return_val[int(not match)][key] = d[key]
    return (*return_val,)


def groupby_prefix_and_trim(prefix, d):
    kwargs_with_prefix, kwargs = group_dict_by_key(lambda k: k.startswith(prefix), d)
    kwargs_without_prefix = dict(map(lambda x: (x[0][len(prefix):], x[1]), tuple(kwargs_with_prefix.items())))
    return kwargs_without_prefix, kwargs


def l2norm(t):
    return F.normalize(t, dim=-1)


def leaky_relu(p=0.1):
    return nn.LeakyReLU(0.1)           # slope is always 0.1 in the reference (vq.py:94-95)


def stable_softmax(t, dim=-1, alpha=32 ** 2):
    t = t / alpha
    t = t - torch.amax(t, dim=dim, keepdim=True).detach()
    return (t * alpha).softmax(dim=dim)


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
        embed = torch.randn(codebook_size, codebook_dim)
        self.register_buffer('embed', l2norm(embed) if use_cosine_sim else embed)
        self.register_buffer('cluster_size', torch.zeros(codebook_size))
        self.register_buffer('initted', torch.tensor([not kmeans_init]))

    @property
    def codebook(self):
        return self.embed

    def forward(self, x):
        if self.accept_image_fmap:
            B, Cc, Hh, Ww = x.shape
            x = x.permute(0, 2, 3, 1).reshape(B, Hh * Ww, Cc)
        x = self.project_in(x)
        flat = x.reshape(-1, x.shape[-1])
        if self.use_cosine_sim:
            fn = l2norm(flat)
            if self.training and not bool(self.initted):
                perm = torch.randint(0, fn.shape[0], (self.embed.shape[0],), device=fn.device)
                self.embed.copy_(fn[perm].detach())
                self.initted.fill_(True)
            sim = fn @ l2norm(self.embed).t()
        else:
            fn = flat
            sim = -torch.cdist(flat, self.embed)
        ind = sim.argmax(dim=-1)
        quant = self.embed[ind].reshape(x.shape)
        loss = torch.zeros(1, device=x.device)
        if self.training:
            onehot = F.one_hot(ind, self.embed.shape[0]).type(flat.dtype)
            self.cluster_size.mul_(self.decay).add_(onehot.sum(0), alpha=1 - self.decay)
            emb_sum = onehot.t() @ fn.detach()
            hit = (onehot.sum(0) > 0)[:, None]
            new = torch.where(hit, l2norm(emb_sum) if self.use_cosine_sim else emb_sum / onehot.sum(0).clamp(min=1)[:, None], self.embed)
            upd = self.embed * self.decay + new * (1 - self.decay)
            self.embed.copy_(l2norm(upd) if self.use_cosine_sim else upd)
            loss = F.mse_loss(quant.detach(), x) * self.commitment_weight
            quant = x + (quant - x).detach()
        quant = self.project_out(quant)
        ind = ind.reshape(x.shape[:-1])
        if self.accept_image_fmap:
            quant = quant.reshape(B, Hh, Ww, -1).permute(0, 3, 1, 2)
            ind = ind.reshape(B, Hh, Ww)
        return quant, ind, loss.reshape(1)


class LayerNormChan(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))
        self.b = nn.Parameter(torch.zeros(1, dim, 1, 1))

    def forward(self, x):
        var = torch.var(x, dim=1, unbiased=False, keepdim=True)
        mean = torch.mean(x, dim=1, keepdim=True)
        return (x - mean) / (var + self.eps).sqrt() * self.g + self.b


class ContinuousPositionBias(nn.Module):
    def __init__(self, *, dim, heads, layers=2):
        super().__init__()
        self.net = MList([])
        self.net.append(nn.Sequential(nn.Linear(2, dim), leaky_relu()))
        for _ in range(layers - 1):
            self.net.append(nn.Sequential(nn.Linear(dim, dim), leaky_relu()))
        self.net.append(nn.Linear(dim, heads))
        self.register_buffer('rel_pos', None, persistent=False)

    def forward(self, x):
        n, device = x.shape[-1], x.device
        fmap_size = int(sqrt(n))
        if not exists(self.rel_pos):
            pos = torch.arange(fmap_size, device=device)
            grid = torch.stack(torch.meshgrid(pos, pos, indexing='ij')).reshape(2, -1).t()
            rel_pos = grid[:, None, :] - grid[None, :, :]
            rel_pos = torch.sign(rel_pos) * torch.log(rel_pos.abs() + 1)
            self.register_buffer('rel_pos', rel_pos, persistent=False)
        rel_pos = self.rel_pos.float()
        for layer in self.net:
            rel_pos = layer(rel_pos)
        return x + rel_pos.permute(2, 0, 1)


class GLUResBlock(nn.Module):
    def __init__(self, chan, groups=16):
        super().__init__()
        self.net = nn.Sequential(
            nn.Conv2d(chan, chan * 2, 3, padding=1), nn.GLU(dim=1), nn.GroupNorm(groups, chan),
            nn.Conv2d(chan, chan * 2, 3, padding=1), nn.GLU(dim=1), nn.GroupNorm(groups, chan),
            nn.Conv2d(chan, chan, 1))

    def forward(self, x):
        return self.net(x) + x


class ResBlock(nn.Module):
    def __init__(self, chan, groups=16):
        super().__init__()
        self.net = nn.Sequential(
            nn.Conv2d(chan, chan, 3, padding=1), nn.GroupNorm(groups, chan), leaky_relu(),
            nn.Conv2d(chan, chan, 3, padding=1), nn.GroupNorm(groups, chan), leaky_relu(),
            nn.Conv2d(chan, chan, 1))

    def forward(self, x):
        return self.net(x) + x


class VQGanAttention(nn.Module):
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
        h = self.heads
        B, _, height, width = x.shape
        residual = x.clone()
        q, k, v = self.to_qkv(x).chunk(3, dim=1)
        q, k, v = map(lambda t: t.reshape(B, h, -1, height * width), (q, k, v))
        q, k = map(l2norm, (q, k))          # over the SPATIAL axis (quirk Q9)
        sim = einsum('b h c i, b h c j -> b h i j', q, k) * self.scale.exp()
        sim = self.cpb(sim)
        attn = self.dropout(stable_softmax(sim, dim=-1))
        out = einsum('b h i j, b h c j -> b h c i', attn, v).reshape(B, -1, height, width)
        out = self.to_out(out)
        return self.post_norm(out) + residual


class VQGanVAE(nn.Module):
    """vq.py:288-548 (GAN/VGG branch excluded)."""

    def __init__(self, *, dim, image_size, channels=3, num_layers=4, layer_mults=None, l2_recon_loss=False,
                 use_hinge_loss=True, num_resnet_blocks=1, vgg=None, vq_codebook_dim=256, vq_codebook_size=512,
                 vq_decay=0.8, vq_commitment_weight=1., vq_kmeans_init=True, vq_use_cosine_sim=True, use_attn=True,
                 attn_dim_head=64, attn_heads=8, resnet_groups=16, attn_dropout=0., first_conv_kernel_size=5,
                 use_vgg_and_gan=True, **kwargs):
        super().__init__()
        assert dim % resnet_groups == 0, f'dimension {dim} must be divisible by {resnet_groups} (groups for the groupnorm)'
        if use_vgg_and_gan:
            raise NotImplementedError('VQGanVAE GAN/VGG training (use_vgg_and_gan=True) is outside the accelerated path; '
                                      'construct with use_vgg_and_gan=False (NUWA uses the VAE frozen)')
        vq_kwargs, kwargs = groupby_prefix_and_trim('vq_', kwargs)
        self.image_size = image_size
        self.channels = channels
        self.num_layers = num_layers
        self.fmap_size = image_size // (num_layers ** 2)        # as in the reference (quirk Q6)
        self.codebook_size = vq_codebook_size
        self.encoders = MList([])
        self.decoders = MList([])
        layer_mults = default(layer_mults, list(map(lambda t: 2 ** t, range(num_layers))))
        assert len(layer_mults) == num_layers, 'layer multipliers must be equal to designated number of layers'
        layer_dims = [dim * mult for mult in layer_mults]
        dims = (dim, *layer_dims)
        dim_pairs = zip(dims[:-1], dims[1:])
        append = lambda arr, t: arr.append(t)
        prepend = lambda arr, t: arr.insert(0, t)
        if not isinstance(num_resnet_blocks, tuple):
            num_resnet_blocks = (*((0,) * (num_layers - 1)), num_resnet_blocks)
        if not isinstance(use_attn, tuple):
            use_attn = (*((False,) * (num_layers - 1)), use_attn)
        assert len(num_resnet_blocks) == num_layers and len(use_attn) == num_layers
        for layer_index, (dim_in, dim_out), layer_num_resnet_blocks, layer_use_attn in zip(range(num_layers), dim_pairs, num_resnet_blocks, use_attn):
            append(self.encoders, nn.Sequential(nn.Conv2d(dim_in, dim_out, 4, stride=2, padding=1), leaky_relu()))
            prepend(self.decoders, nn.Sequential(nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False), nn.Conv2d(dim_out, dim_in, 3, padding=1), leaky_relu()))
            if layer_use_attn:
                prepend(self.decoders, VQGanAttention(dim=dim_out, heads=attn_heads, dim_head=attn_dim_head, dropout=attn_dropout))
            for _ in range(layer_num_resnet_blocks):
                append(self.encoders, ResBlock(dim_out, groups=resnet_groups))
                prepend(self.decoders, GLUResBlock(dim_out, groups=resnet_groups))
            if layer_use_attn:
                append(self.encoders, VQGanAttention(dim=dim_out, heads=attn_heads, dim_head=attn_dim_head, dropout=attn_dropout))
        prepend(self.encoders, nn.Conv2d(channels, dim, first_conv_kernel_size, padding=first_conv_kernel_size // 2))
        append(self.decoders, nn.Conv2d(dim, channels, 1))
        self.vq = VectorQuantize(dim=layer_dims[-1], codebook_dim=vq_codebook_dim, codebook_size=vq_codebook_size,
                                 decay=vq_decay, commitment_weight=vq_commitment_weight, accept_image_fmap=True,
                                 kmeans_init=vq_kmeans_init, use_cosine_sim=vq_use_cosine_sim, **vq_kwargs)
        self.recon_loss_fn = F.mse_loss if l2_recon_loss else F.l1_loss
        self.vgg = None
        self.discr = None
        self.use_vgg_and_gan = use_vgg_and_gan

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
        indices = self._hip_encode_indices(images)
        return indices.reshape(b, f, *indices.shape[1:])

    def forward(self, img, return_loss=False, return_discr_loss=False, return_recons=False, apply_grad_penalty=False):
        batch, channels, height, width = img.shape
        assert height == self.image_size and width == self.image_size, 'height and width of input image must be equal to {self.image_size}'
        assert channels == self.channels, 'number of channels on image or sketch is not equal to the channels set on this VQGanVAE'
        fmap, indices, commit_loss = self.encode(img)
        fmap = self.decode(fmap)
        if not return_loss and not return_discr_loss:
            return fmap
        assert not return_discr_loss, 'discriminator training is outside the accelerated path'
        recon_loss = self.recon_loss_fn(fmap, img)        # only term when use_vgg_and_gan=False (quirk Q17)
        if return_recons:
            return recon_loss, fmap
        return recon_loss
