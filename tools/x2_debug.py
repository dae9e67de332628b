import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K
torch.manual_seed(33)
B, n, T, heads, dh = 1, 64, 256, 8, 64
inner = heads * dh
dev = 'cuda'
bf = lambda *s: K.BF(torch.randn(*s, device=dev).to(torch.bfloat16), None)
q, kv, do = bf(B * n, inner), bf(B * T, 2 * inner), bf(B * n, inner)
nk, nv = torch.randn(heads, dh, device=dev), torch.randn(heads, dh, device=dev)
wth = (torch.randn(heads, heads, device=dev) * 0.5 + torch.eye(heads, device=dev)).contiguous()
for full_mask in (False, True):
    mask = torch.rand(B, T, device=dev) > 0.3
    if full_mask:
        mask[0] = False
    g = K.x_geom(B, n, T, heads, dh)
    pk = K.xattn_pack(g, kv, nk, nv, mask.to(torch.uint8))
    o, P, Pm = K.xattn_fwd(g, q, pk, wth)
    dq, dS, dwth = K.xattn_bwd(g, do, pk, wth, P)
    o2, stats = K.xattn2_fwd(g, q, pk, wth)
    dq2, dS2, Pm2, dwth2 = K.xattn2_bwd(g, q, do, pk, wth, stats)
    torch.cuda.synchronize()
    print('full_mask', full_mask, 'stats finite', bool(torch.isfinite(stats).all()), 'stats min/max', float(stats.min()), float(stats.max()))
    for nm, a, b in (('o', o2.hi, o.hi), ('dq', dq2.hi, dq.hi), ('dS', dS2.hi, dS.hi), ('Pm', Pm2.hi, Pm.hi), ('dwth', dwth2, dwth)):
        a, b = a.float(), b.float()
        d = (a - b).abs()
        print(f'  {nm}: max|new| {float(a.abs().max()):.3e} max|old| {float(b.abs().max()):.3e} maxdiff {float(d.max()):.3e} argmax {tuple(int(v) for v in torch.unravel_index(d.argmax(), d.shape))} nbad {int((d > 1e-1 * b.abs().max()).sum())}')
    bad = (dS2.hi.float() - dS.hi.float()).abs() > 0.1
    if bad.any():
        idx = bad.nonzero()
        print('  dS bad heads', sorted(set(idx[:, 1].tolist())), 'queries', sorted(set(idx[:, 2].tolist()))[:20], 'keys', sorted(set(idx[:, 3].tolist()))[:40])
