import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K
B, n, T, heads, dh = 1, 64, 256, 8, 64
inner = 512
torch.manual_seed(13)
dev = 'cuda'
q = K.BF(torch.randn(B * n, inner, device=dev).to(torch.bfloat16), None)
do = K.BF(torch.randn(B * n, inner, device=dev).to(torch.bfloat16), None)
kv = K.BF(torch.randn(B * T, 2 * inner, device=dev).to(torch.bfloat16), None)
nk, nv = torch.randn(heads, dh, device=dev), torch.randn(heads, dh, device=dev)
wth = (torch.randn(heads, heads, device=dev) * 0.5 + torch.eye(heads, device=dev)).contiguous()
mask = (torch.rand(B, T, device=dev) > 0.3).to(torch.uint8); mask[0] = 0
g = K.x_geom(B, n, T, heads, dh)
pk6 = K.xattn6_pack(g, kv.hi, mask)
o, stats = K.xattn6_fwd(g, q.hi, pk6, nk, nv, wth, lo=False)
pkb = K.xattn6_pack_bwd(g, kv.hi, nk, nv, mask)
pko = K.xattn_pack(g, kv, nk, nv, mask)
print('K image equal to old image:', torch.equal(pkb.K6.permute(0, 2, 1, 3, 4).reshape(B, heads, g.JP, dh)[..., :0], pko.Kp.hi[..., :0]))
dq, dS, Pm, dwth = K.xattn6_bwd(g, q, do, pkb, wth, stats)
dq2, dS2, Pm2, dwth2 = K.xattn2_bwd(g, q, do, pko, wth, stats, chunk_major=True)
torch.cuda.synchronize()
for name, a, b in (('dq', dq.hi, dq2.hi), ('dS', dS.hi, dS2.hi), ('Pm', Pm.hi, Pm2.hi), ('dwth', dwth, dwth2)):
    a, b = a.float(), b.float()
    bad = ~torch.isfinite(a)
    d = (a - b).abs()
    d[~torch.isfinite(d)] = 1e30
    print(name, 'shape', tuple(a.shape), 'nonfinite', int(bad.sum()), 'max|a|', a[~bad].abs().max().item(), 'max|b|', b.abs().max().item(), 'max diff', d.max().item(), 'argmax', int(d.argmax()))
mx = K.xattn_permuted_extent(g)
r6, r2 = K.xattn_rows(g, Pm.hi).float()[..., :mx], K.xattn_rows(g, Pm2.hi).float()[..., :mx]
dd = (r6 - r2).abs()
print('Pm rows diff max', dd.max().item(), 'per head', dd.amax(dim=(0, 2, 3)).tolist())
r6, r2 = K.xattn_rows(g, dS.hi).float()[..., :mx], K.xattn_rows(g, dS2.hi).float()[..., :mx]
dd = (r6 - r2).abs()
print('dS rows diff max', dd.max().item(), 'per head', dd.amax(dim=(0, 2, 3)).tolist(), 'per chunk', dd.reshape(B, heads, n, -1, 8).amax(dim=(0, 1, 2, 4)).tolist()[:40])
dqd = (dq.hi.float() - dq2.hi.float()).abs().reshape(n, heads, dh)
print('dq diff per head', dqd.amax(dim=(0, 2)).tolist())
print('dq diff per query (first 16)', dqd.amax(dim=(1, 2))[:16].tolist())
