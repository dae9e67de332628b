#!/usr/bin/env python
"""which gradients of the 3-layer cfg-3 bench step differ between runs of the same step, per fp16-gradient class set (AMDNUWA_BWD_F16)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import nuwa_pytorch_amd as A
from nuwa_pytorch_amd import kernels as K

DEV = 'cuda'
c = dict(bench.CFGS['cfg3'], dec_depth=3)
nuwa = bench.build_model(c, DEV)
names = [n for n, _ in nuwa.named_parameters()]
params = bench.decoder_params(nuwa)
pn = {id(p): n for n, p in nuwa.named_parameters()}
g = torch.Generator().manual_seed(5)
b, N = int(os.environ.get('B', '16')), c['frames'] * c['fmap'] ** 2
ids = torch.randint(0, c['codebook'], (b, N), generator=g).to(DEV)
ctx = torch.randn(b, c['text_len'], c['dim'], generator=g).to(DEV)
mask = (torch.rand(b, c['text_len'], generator=g) > 0.1).to(DEV)
A.set_precision('bf16x3-fwd')
for cls in sys.argv[1:] or ['f', 'fs', 'fx', 'fsx']:
    K.set_bwd_f16(cls)
    runs = []
    for _ in range(4):
        for p in params:
            p.grad = None
        loss = bench.decoder_step(nuwa, ids, ctx, mask)
        torch.cuda.synchronize()
        runs.append([p.grad.detach().clone() for p in params])
    bad = set()
    for k in range(1, 4):
        for i, (x, y) in enumerate(zip(runs[0], runs[k])):
            if not torch.equal(x, y):
                bad.add(i)
    print(f'classes {cls!r}: {len(bad)} of {len(params)} gradients differ:', [pn[id(params[i])] for i in sorted(bad)][:40], flush=True)

# where do non-finite values appear (classes 'fx')?  amax of the fp16 gradients around every cross-attention backward of one step
K.set_bwd_f16('fx')
orig = K.xattn6_bwd16
def spy(g_, q16, dO16, pk, wth, stats, s2):
    r = orig(g_, q16, dO16, pk, wth, stats, s2)
    dq, dS, Pm, dwth = r
    f = lambda t: (float(t.float().abs().nan_to_num(nan=-1.0, posinf=-2.0).max()), int((~torch.isfinite(t.float())).sum()))
    print('xattn6_bwd16: S', float(s2[0]), 'dO', f(dO16), 'dq', f(dq), 'dS', f(dS), 'Pm', f(Pm), 'dwth', f(dwth), flush=True)
    return r
K.xattn6_bwd16 = spy
for p in params:
    p.grad = None
bench.decoder_step(nuwa, ids, ctx, mask)
torch.cuda.synchronize()
print('non-finite gradients:', [pn[id(p)] for p in params if not bool(torch.isfinite(p.grad).all())])
