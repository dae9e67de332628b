#!/usr/bin/env python
"""Bit-reproducibility of the attention kernels at batches that fill the chip (several workgroups per CU, several rounds): every kernel is
run 8 times on the same inputs and every output compared with the first run's, element for element.  The full-size determinism tests of
the GPU suite use one sample (one workgroup per CU at most); a race that needs co-resident workgroups only shows up here."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuwa_pytorch_amd import kernels as K, _lib  # noqa: E402

L = _lib.lib()
DEV = 'cuda'
heads, dh = 8, 64
inner = heads * dh
REP = 8


def tensors(o):
    out = []
    for x in (o if isinstance(o, (tuple, list)) else (o,)):
        if isinstance(x, K.BF):
            out += [t for t in (x.hi, x.lo, x.f16) if t is not None]
        elif torch.is_tensor(x):
            out.append(x)
    return out


def check(name, fn):
    ref = [t.clone() for t in tensors(fn())]
    bad = 0
    for _ in range(REP - 1):
        for a, b in zip(tensors(fn()), ref):
            bad += int((a != b).sum()) if a.dtype != torch.float32 else int((a.view(torch.int32) != b.view(torch.int32)).sum())
    print(f'{name:70s} {"bit-identical" if bad == 0 else f"{bad} ELEMENTS DIFFER"} over {REP} runs', flush=True)
    return bad


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
    only_s3 = '--only-s3' in sys.argv           # (the build-variant experiment of round 5: the 3DNA kernels alone)
    print('library:', _lib.LIB_PATH, flush=True)
    torch.manual_seed(0)
    total = 0
    wth = (torch.randn(heads, heads) * 0.5 + torch.eye(heads)).to(DEV)
    shape, kern = (10, 16, 16), (5, 3, 3)
    n = 2560
    for dil in ((1, 1, 1), (2, 2, 2), (4, 4, 4)):
        qkv = torch.randn(B * n, 3 * inner, device=DEV)
        g = K.s3_geom(B, n, shape, kern, dil, heads, dh)
        pbf = K.BF(qkv.to(torch.bfloat16), None)
        p16 = K.BF(qkv.to(torch.bfloat16), None, qkv.half())
        dO = K.BF(torch.randn(B * n, inner, device=DEV).to(torch.bfloat16), None)
        for rows in (1, 2):
            L.amdnuwa_set_tuning(16, rows)
            total += check(f'3DNA fwd bf16, dilation {dil[0]}, {rows} row(s) per tile, b={B}', lambda: K.sparse3dna_fwd(g, pbf, wth))
            total += check(f'3DNA fwd fp16, dilation {dil[0]}, {rows} row(s) per tile, b={B}', lambda: K.sparse3dna_fwd(g, p16, wth))
        L.amdnuwa_set_tuning(16, 0)
        total += check(f'3DNA bwd bf16, dilation {dil[0]}, b={B}', lambda: K.sparse3dna_bwd(g, pbf, wth, dO))
    if only_s3:
        print('TOTAL differing elements:', total)
        return 1 if total else 0
    T = 256
    q = torch.randn(B * n, inner, device=DEV)
    kv = torch.randn(B * T, 2 * inner, device=DEV)
    gx = K.x_geom(B, n, T, heads, dh)
    mask = (torch.rand(B, T, device=DEV) > 0.2).to(torch.uint8)
    nk, nv = torch.randn(heads, dh, device=DEV), torch.randn(heads, dh, device=DEV)
    q16 = K.BF(q.to(torch.bfloat16), None, q.half())
    kv16 = K.BF(kv.to(torch.bfloat16), None, kv.half())
    pk16 = K.xattn_pack(gx, kv16, nk, nv, mask)
    total += check(f'cross attention fwd fp16 (xattn4), b={B}', lambda: K.xattn2_fwd_f16(gx, q16, pk16, wth))
    qb = K.BF(q.to(torch.bfloat16), None)
    pkb = K.xattn_pack(gx, K.BF(kv.to(torch.bfloat16), None), nk, nv, mask)
    total += check(f'cross attention fwd bf16 (xattn4), b={B}', lambda: K.xattn2_fwd(gx, qb, pkb, wth))
    o, stats = K.xattn2_fwd(gx, qb, pkb, wth)
    dO = K.BF(torch.randn(B * n, inner, device=DEV).to(torch.bfloat16), None)
    total += check(f'cross attention bwd (xattn3, query side), b={B}', lambda: K.xattn2_bwd(gx, qb, dO, pkb, wth, stats))
    print('TOTAL differing elements:', total)
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main())
