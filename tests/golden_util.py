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
