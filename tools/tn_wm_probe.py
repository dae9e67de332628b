#!/usr/bin/env python
"""The batched dK / dV products of the cross attention (fp16 chunk-major dS / P', b * heads batch elements of [264 x 64] over 2560 token rows):
the lean whole-M kernel (gemm_tn_wmf_kernel, default) against the general one (gemm_tn_wm_kernel, tuning key 25 = 3); results must be bit-identical."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from xattn6_bench import bench  # noqa: E402


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = 'cuda'
    n, heads, dh, T = 2560, 8, 64, 256
    inner = heads * dh
    K.set_precision('bf16x3-fwd')
    L = _lib.lib()
    torch.manual_seed(0)
    g = K.x_geom(b, n, T, heads, dh)
    q16 = torch.randn(b * n, inner, device=dev).half()
    do16 = torch.randn(b * n, inner, device=dev).half()
    s2 = torch.tensor([1.0, 1.0], device=dev)
    shape = (g.B, g.heads, g.JP // 32, g.n, 32)
    dS16 = torch.randn(shape, device=dev).half()
    Pm16 = torch.randn(shape, device=dev).half()
    Mx = K.xattn_permuted_extent(g)
    ref = None
    for rnd in range(3):
        row = []
        for key in (3, 0):
            L.amdnuwa_set_tuning(25, key)
            t = bench(lambda: K.xattn_kv_grads16(g, dS16, Pm16, q16, do16, s2), 10)
            dk, dv = K.xattn_kv_grads16(g, dS16, Pm16, q16, do16, s2)
            if ref is None:
                ref = (dk[:, :, :Mx].clone(), dv[:, :, :Mx].clone())
            same = bool(torch.equal(dk[:, :, :Mx], ref[0]) and torch.equal(dv[:, :, :Mx], ref[1]))
            row.append(f'{"general" if key == 3 else "lean   "}: {t:7.1f} us for both products, {"same bits" if same else "DIFFERENT"} ')
        L.amdnuwa_set_tuning(25, 0)
        print(' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
