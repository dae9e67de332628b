import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    """returns (arrays dict of torch tensors, params dict, grads dict) for tests/golden/<name>.npz"""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    A, P, G = {}, {}, {}
    for k in z.files:
        v = torch.from_numpy(z[k])
        if k.startswith('p.'):
            P[k[2:]] = v
        elif k.startswith('g.'):
            G[k[2:]] = v
        else:
            A[k] = v
    return A, P, G


def load_raw(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def tup(t):
    return tuple(int(v) for v in t.reshape(-1))


def rel_err(a, b):
    """max-abs error relative to the reference's max-abs (the metric the tests quote)"""
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))


def fill_params(module, seed=0):
    """deterministic, construction-order-independent parameter / buffer values: every state-dict entry is drawn from a generator
    seeded by (its name, seed), so the reference module (in the build container) and the product module (anywhere) end up with
    identical weights without a multi-megabyte fixture.  VQ buffer names are normalised (`vq._codebook.embed` == `vq.embed`)."""
    import zlib
    with torch.no_grad():
        for name, t in module.state_dict().items():
            key = name.replace('._codebook.', '.')
            g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
            if not t.is_floating_point():
                if t.dtype == torch.bool:
                    t.fill_(True)                                   # `initted`: codebook in use
                continue
            leaf = key.rsplit('.', 1)[-1]
            if leaf == 'embed':                                     # codebook rows: unit vectors
                t.copy_(torch.nn.functional.normalize(torch.randn(t.shape, generator=g), dim=-1))
            elif leaf == 'cluster_size':
                t.zero_()
            elif t.dim() <= 1 or leaf in ('g', 'b', 'scale', 'bias'):
                base = 1.0 if leaf in ('weight', 'g') else (float(t.reshape(-1)[0]) if leaf == 'scale' else 0.0)
                t.copy_(base + 0.1 * torch.randn(t.shape, generator=g))
            else:
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * fan_in ** -0.5)
    return module


def sample2048(t):
    """<= 2048 evenly strided elements of a tensor (what the g12 fixture keeps of large gradients)"""
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // 2048)][:2048].clone()
