"""
ORACLE -- TEST INFRASTRUCTURE ONLY (build container only).

Lets the read-only reference checkout at /root/reference be imported in the build container,
whose image lacks four of the reference's third-party dependencies.  Nothing here travels to
the GPU box as a dependency of the product; `/root/reference` does not exist there and every
user of this module skips when it is absent.

  unfoldNd                -> exact gather shim (pure index arithmetic; bit-identical to F.unfold
                             on 2-D inputs, checked in tests/test_oracle_vs_reference.py)
  vector_quantize_pytorch -> restatement of the documented cosine-sim VectorQuantize
                             (PARITY UNPINNED: upstream source is not available offline)
  torchvision, ftfy       -> empty modules (import-time only; never executed on the hot path)
"""
import os
import sys
import types

import torch
import torch.nn.functional as F
from torch import nn

REFERENCE_ROOT = os.environ.get('NUWA_REFERENCE_ROOT', '/root/reference')


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'nuwa_pytorch'))


def unfoldNd(input, kernel_size, dilation=1, padding=0, stride=1):
    """N-d im2col: (B, C, *spatial) -> (B, C*prod(k), L); channel-major, taps row-major,
    output positions row-major.  stride 1 / padding 0 only (all the reference uses:
    np.py:447, 526, 662)."""
    nd = input.dim() - 2
    ks = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size,) * nd
    ds = dilation if isinstance(dilation, (tuple, list)) else (dilation,) * nd
    assert padding == 0 and stride == 1
    x = input
    for i, (k, d) in enumerate(zip(ks, ds)):
        span = (k - 1) * d + 1
        x = x.unfold(2 + i, span, 1)          # appends a window dim at the end
        x = x[..., ::d]
    # x: (B, C, *out, *k)
    B, C = x.shape[:2]
    out_dims = x.shape[2:2 + nd]
    perm = [0, 1] + list(range(2 + nd, 2 + 2 * nd)) + list(range(2, 2 + nd))
    x = x.permute(*perm).reshape(B, C * int(torch.tensor(ks).prod()), int(torch.tensor(out_dims).prod()))
    return x


class VectorQuantize(nn.Module):
    """Restatement of vector_quantize_pytorch.VectorQuantize for the kwargs the reference
    passes (vq.py:368-378).  PARITY UNPINNED."""

    def __init__(self, dim, codebook_size, codebook_dim=None, decay=0.8, commitment_weight=1.,
                 accept_image_fmap=False, kmeans_init=False, use_cosine_sim=False, eps=1e-5, **kwargs):
        super().__init__()
        codebook_dim = codebook_dim if codebook_dim is not None else dim
        self.project_in = nn.Linear(dim, codebook_dim) if codebook_dim != dim else nn.Identity()
        self.project_out = nn.Linear(codebook_dim, dim) if codebook_dim != dim else nn.Identity()
        self.decay, self.commitment_weight, self.eps = decay, commitment_weight, eps
        self.accept_image_fmap = accept_image_fmap
        self.use_cosine_sim = use_cosine_sim
        embed = F.normalize(torch.randn(codebook_size, codebook_dim), dim=-1) if use_cosine_sim \
            else torch.randn(codebook_size, codebook_dim)
        self.register_buffer('embed', embed)
        self.register_buffer('cluster_size', torch.zeros(codebook_size))
        self.register_buffer('initted', torch.tensor([not kmeans_init]))

    @property
    def codebook(self):
        return self.embed

    def forward(self, x):
        if self.accept_image_fmap:
            B, C, Hh, Ww = x.shape
            x = x.permute(0, 2, 3, 1).reshape(B, Hh * Ww, C)
        x = self.project_in(x)
        flat = x.reshape(-1, x.shape[-1])
        if self.use_cosine_sim:
            fn = F.normalize(flat, dim=-1)
            if self.training and not bool(self.initted):
                perm = torch.randperm(fn.shape[0])[:self.embed.shape[0]]
                if perm.numel() < self.embed.shape[0]:
                    perm = torch.randint(0, fn.shape[0], (self.embed.shape[0],))
                self.embed.copy_(fn[perm].detach())
                self.initted.fill_(True)
            sim = fn @ F.normalize(self.embed, dim=-1).t()
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
            mask = (onehot.sum(0) > 0)[:, None]
            new = torch.where(mask, F.normalize(emb_sum, dim=-1), self.embed)
            self.embed.copy_(F.normalize(self.embed * self.decay + new * (1 - self.decay), dim=-1)
                             if self.use_cosine_sim else self.embed * self.decay + new * (1 - self.decay))
            loss = F.mse_loss(quant.detach(), x) * self.commitment_weight
            quant = x + (quant - x).detach()
        quant = self.project_out(quant)
        ind = ind.reshape(x.shape[:-1])
        if self.accept_image_fmap:
            quant = quant.reshape(B, Hh, Ww, -1).permute(0, 3, 1, 2)
            ind = ind.reshape(B, Hh, Ww)
        return quant, ind, loss.reshape(1)


def install():
    """register the stub modules and put the reference on sys.path; returns the imported
    `nuwa_pytorch` reference package."""
    if not reference_available():
        raise RuntimeError('reference checkout not present')
    m = types.ModuleType('unfoldNd'); m.unfoldNd = unfoldNd
    sys.modules.setdefault('unfoldNd', m)
    m = types.ModuleType('vector_quantize_pytorch'); m.VectorQuantize = VectorQuantize
    sys.modules.setdefault('vector_quantize_pytorch', m)
    if 'torchvision' not in sys.modules:
        tv = types.ModuleType('torchvision')
        for subname in ('transforms', 'utils', 'datasets', 'models'):
            sm = types.ModuleType('torchvision.' + subname)
            setattr(tv, subname, sm)
            sys.modules['torchvision.' + subname] = sm
        tv.transforms.Compose = tv.transforms.Lambda = tv.transforms.Resize = object
        tv.utils.make_grid = tv.utils.save_image = lambda *a, **k: None
        tv.datasets.ImageFolder = object
        sys.modules['torchvision'] = tv
        T = tv.transforms
        for n in ('RandomHorizontalFlip', 'CenterCrop', 'ToTensor', 'ToPILImage', 'RandomCrop', 'Normalize'):
            setattr(T, n, object)
    if 'ftfy' not in sys.modules:
        ft = types.ModuleType('ftfy'); ft.fix_text = lambda s: s
        sys.modules['ftfy'] = ft
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import nuwa_pytorch  # noqa: the reference package
    return nuwa_pytorch
