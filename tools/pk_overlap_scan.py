#!/usr/bin/env python
"""Scan a gfx950 assembly listing for packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose DESTINATION pair is
also a SOURCE pair read ACROSS halves: the low result reads the source's high register or the high result reads its low register (op_sel /
op_sel_hi).  If the two halves of such an instruction are not executed from operands read up front, the half computed second sees the other
half's result instead of the operand.  This is the shape of the instructions in the head-mix loop of s3_fwd_tile_kernel<2> that stored wrong
low-half results under LDS back-pressure (round 4): a HYPOTHESIS about the mechanism, kept as a tool so that the pattern can
be counted per kernel.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only nuwa_pytorch_amd/csrc/sparse3dna.hip -o /tmp/s3.s
    python tools/pk_overlap_scan.py /tmp/s3.s"""
import re
import sys


def pair(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    return (int(m.group(1)), int(m.group(2))) if m else None


def sel(t, name, n, default):
    m = re.search(name + r':\[([01,]+)\]', t)
    if not m:
        return [default] * n
    v = [int(x) for x in m.group(1).split(',')]
    return v + [default] * (n - len(v))


def main():
    path = sys.argv[1]
    kernel, hits, total = None, {}, {}
    for raw in open(path):
        t = raw.strip()
        m = re.match(r'^(_Z\w+):', t)
        if m:
            kernel = m.group(1)
        op = t.split()[0] if t else ''
        if op not in ('v_pk_fma_f32', 'v_pk_mul_f32', 'v_pk_add_f32'):
            continue
        total[kernel] = total.get(kernel, 0) + 1
        body = re.split(r'\s+op_sel', t[len(op):])[0]
        ops = [x.strip() for x in body.split(',')]
        dst = pair(ops[0])
        srcs = ops[1:]
        lo_sel, hi_sel = sel(t, 'op_sel', len(srcs), 0), sel(t, 'op_sel_hi', len(srcs), 1)
        for k, sname in enumerate(srcs):
            if dst and pair(sname) == dst and (lo_sel[k] == 1 or hi_sel[k] == 0):
                hits.setdefault(kernel, []).append(t)
                break
    for k in sorted(total, key=lambda x: -len(hits.get(x, []))):
        h = hits.get(k, [])
        if h:
            print(f'{len(h):5d} of {total[k]:5d} packed fp32 ops overwrite a source pair they read across halves: {k}')
            print(f'         e.g. {h[0]}')
    print('kernels with packed fp32 ops:', len(total), '; with the overlap pattern:', len(hits))
    return 0


if __name__ == '__main__':
    sys.exit(main())
