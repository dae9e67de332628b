import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nuwa_pytorch_amd import kernels as K, _lib
from gemm_bench import bench
L = _lib.lib()
b, n, heads, dh = 128, 2560, 8, 64
inner = heads * dh
mk = lambda c: K.BF((torch.randn(b * n, c, device='cuda') * 0.5).to(torch.bfloat16), None)
qkv, do = mk(3 * inner), mk(inner)
wth = (torch.randn(heads, heads) * 0.3 + torch.eye(heads)).cuda()
for dil in (1, 2, 4):
    g = K.s3_geom(b, n, (10, 16, 16), (5, 3, 3), (dil, dil, dil), heads, dh)
    row = []
    for name, key3 in (('f-major', 2), ('y-major', 0)):
        L.amdnuwa_set_tuning(3, key3)
        tf = bench(lambda: K.sparse3dna_fwd(g, qkv, wth), 5)
        tb = bench(lambda: K.sparse3dna_bwd(g, qkv, wth, do), 5)
        row.append(f'{name}: fwd {tf*1e6:7.1f} bwd {tb*1e6:7.1f}')
    L.amdnuwa_set_tuning(3, 0)
    print(f'dilation {dil}: ' + ' | '.join(row))
