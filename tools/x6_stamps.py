#!/usr/bin/env python
"""Per-phase s_memtime stamps of the xattn6 forward (build variant x6t: python -m nuwa_pytorch_amd.build --variant x6t; run with
AMDNUWA_LIBRARY=nuwa_pytorch_amd/lib_x6t/libamdnuwa.so): workgroup 0, first item, every wave, every ring step."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402

L = _lib.lib()
b, n, heads, dh, T = 128, 2560, 8, 64, 256
g = K.x_geom(b, n, T, heads, dh)
q16 = torch.randn(b * n, 512, device='cuda').half(); kv16 = torch.randn(b * T, 1024, device='cuda').half()
nk, nv = torch.randn(8, 64, device='cuda'), torch.randn(8, 64, device='cuda')
wth = (torch.randn(8, 8, device='cuda') * 0.3 + torch.eye(8, device='cuda')).contiguous()
mask = (torch.rand(b, T, device='cuda') > 0.2).to(torch.uint8)
pk = K.xattn6_pack(g, kv16, mask)
st = torch.zeros(8 * 16 * 8, dtype=torch.int64, device='cuda')
raw = C.CDLL(_lib.LIB_PATH)
raw.amdnuwa_xattn6_set_stamps.argtypes = [C.c_void_p]
for _ in range(3):
    K.xattn6_fwd(g, q16, pk, nk, nv, wth, o_f16=True)
assert raw.amdnuwa_xattn6_set_stamps(st.data_ptr()) == 0
K.xattn6_fwd(g, q16, pk, nk, nv, wth, o_f16=True)
torch.cuda.synchronize()
t = st.cpu().reshape(8, 16, 8)
t0 = t[:, 12, 0].min().item()
print('item: step 12 = [item top, q + null key done, prologue vmcnt, barrier, pass 1 done, stats done, pass-2 barrier]; step 13 = [pass 2 done, null key done, stores issued]')
print('s_memtime ticks (100 MHz?) relative to the first stamp; pass 1 steps 0-3: [top, done, vmcnt, barrier]; pass 2 steps 4-11: [top, puts issued, lgkm, barrier A, mix+PV issued, vmcnt, barrier B]')
for w in (0, 1, 6, 7):
    print(f'wave {w}')
    for s_ in (12, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13):
        row = t[w, s_]
        print(f'  step {s_:2d}: ' + ' '.join(f'{(x.item() - t0):8d}' if x.item() else '       -' for x in row[:7]))
