#!/usr/bin/env python
"""Cross-attention backward at cfg-3 size: the query-side kernel with and without its dS / Pm stores (tuning key 10 bit 4: probe only),
the two batched TN GEMMs that read them back -- what the dS / P' round trip through HBM costs -- and the recomputing form
(amdnuwa_xattn2_bwd_rc) that replaces both."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402
from gemm_bench import bench  # noqa: E402

L = _lib.lib()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, T, heads, dh = 2560, 256, 8, 64
inner = heads * dh
g = K.x_geom(b, n, T, heads, dh)
mk = lambda r, c: K.BF((torch.randn(r, c, device='cuda') * 0.5).to(torch.bfloat16), None)
q, do, kv = mk(b * n, inner), mk(b * n, inner), mk(b * T, 2 * inner)
nk, nv = torch.randn(heads, dh, device='cuda'), torch.randn(heads, dh, device='cuda')
wth = (torch.randn(heads, heads) * 0.3 + torch.eye(heads)).cuda()
mask = torch.ones(b, T, dtype=torch.uint8, device='cuda')
pk = K.xattn_pack(g, kv, nk, nv, mask)
o, stats = K.xattn2_fwd(g, q, pk, wth)
t_full = bench(lambda: K.xattn2_bwd(g, q, do, pk, wth, stats), 5)
L.amdnuwa_set_tuning(10, 16)
t_nost = bench(lambda: K.xattn2_bwd(g, q, do, pk, wth, stats), 5)
L.amdnuwa_set_tuning(10, 0)
dq, dS, Pm, dwth = K.xattn2_bwd(g, q, do, pk, wth, stats)
t_tn = bench(lambda: K.xattn_kv_grads(g, dS, Pm, q, do), 5)
t_rc = bench(lambda: K.xattn2_bwd_rc(g, q, do, pk, wth, stats), 5)
L.amdnuwa_set_tuning(10, 32)
t_rc_plain = bench(lambda: K.xattn2_bwd_rc(g, q, do, pk, wth, stats), 5)
L.amdnuwa_set_tuning(10, 0)
print(f'b = {b}: recomputing form with the plain block order on the key side {t_rc_plain * 1e6:7.1f} us')
print(f'b = {b}: recomputing form (query side without stores + key-side kernel) {t_rc * 1e6:7.1f} us  vs  {(t_full + t_tn) * 1e6:7.1f} us')
print(f'b = {b}: xattn3_bwd {t_full * 1e6:7.1f} us | without its dS / Pm stores {t_nost * 1e6:7.1f} us | dK / dV TN GEMMs over dS / Pm {t_tn * 1e6:7.1f} us')
