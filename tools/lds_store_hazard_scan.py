#!/usr/bin/env python
"""Scan a gfx950 assembly listing for wide LDS stores (ds_write_b96 / _b128 / ds_write2_b64 / ds_write2st64_b64) whose data registers are
overwritten by one of the next N instructions.  On MI355X such a store can read its data registers after the following VALU instruction has
rewritten them when the LDS queue is backed up (two workgroups per CU): found with tools/determinism_stress.py in the head mix of
s3_fwd_tile_kernel<2> (round 4).  The compiler's hazard recogniser covers this for global stores, not for LDS stores.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only nuwa_pytorch_amd/csrc/sparse3dna.hip -o /tmp/s3.s
    python tools/lds_store_hazard_scan.py /tmp/s3.s [N=6]"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def main():
    path = sys.argv[1]
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    kernel, lines = None, []
    hits = {}
    for raw in open(path):
        t = raw.strip()
        m = re.match(r'^(_Z\w+|\w+):\s*(;.*)?$', t)
        if m and not t.startswith('.'):
            kernel = m.group(1)
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        lines.append((kernel, t))
    for i, (k, t) in enumerate(lines):
        op = t.split()[0]
        if op not in ('ds_write_b128', 'ds_write_b96', 'ds_write2_b64', 'ds_write2st64_b64', 'ds_store_b128'):
            continue
        ops = [x.strip() for x in t[len(op):].split(',')]
        data = set()
        for o in ops[1:]:
            data |= regs(o.split()[0])
        for j in range(1, N + 1):
            if i + j >= len(lines) or lines[i + j][0] != k:
                break
            t2 = lines[i + j][1]
            op2 = t2.split()[0]
            if op2.startswith('s_waitcnt') and 'lgkmcnt(0)' in t2:
                break
            if op2.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_barrier')):
                break
            if op2.startswith('v_') and not op2.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
                dst = regs(t2[len(op2):].split(',')[0].strip())
                if dst & data:
                    hits.setdefault(k, []).append((t, t2, j))
                    break
    for k, hs in hits.items():
        print(f'{k}: {len(hs)} wide LDS store(s) whose data registers are rewritten within {N} instructions')
        for t, t2, j in hs[:4]:
            print(f'    {t}\n        +{j}: {t2}')
    if not hits:
        print('no wide LDS store is followed by a write of its data registers within', N, 'instructions')
    return 1 if hits else 0


if __name__ == '__main__':
    sys.exit(main())
