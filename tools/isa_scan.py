#!/usr/bin/env python
"""Per-kernel instruction-event summary of a gfx950 assembly listing: the tool behind DESIGN.md sections 5g / 5h.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only nuwa_pytorch_amd/csrc/gemm.hip -o /tmp/gemm.s
    python tools/isa_scan.py /tmp/gemm.s gemm_tn_256 [max_chars]

For every kernel whose mangled name contains the pattern it prints the sequence of memory / matrix / synchronisation events in
program-layout order, run-length encoded:
    GL / GS   global (or buffer) load / store          DMA   global_load_lds (LDS-DMA piece)      SCR  scratch access (spill)
    R / W     ds_read* / ds_write*                     DS    other LDS-crossbar ops (bpermute, swizzle)
    M         v_mfma*                                  v     any other VALU instruction           br   branch
    BAR       s_barrier                                [..]  s_waitcnt with its counters          LOOP(label) loop header
What to look for (each of these was found and fixed in round 2):
    GL [vmcnt(0)] GL [vmcnt(0)] ...        loads under branches: the wait-count pass lost the in-order count, nothing overlaps
    DMA ... [vmcnt(0)] R                   the compiler drains an LDS-DMA ring before a fragment read (use dma16_asm, common.h)
    DS [lgkmcnt(0)] v DS [lgkmcnt(0)] ...  dependent cross-lane reductions issued one at a time
    SCR inside a LOOP                      spills on the hot path
Waits that come from inline asm are marked ASM (the compiler does not see those)."""
import re
import sys


def events(lines):
    ev, inasm = [], False
    for l in lines:
        t = l.strip()
        if 'ASMSTART' in l:
            inasm = True
        elif 'ASMEND' in l:
            inasm = False
        elif t.startswith('s_waitcnt'):
            ev.append(('ASM' if inasm else '') + '[' + t.replace('s_waitcnt ', '') + ']')
        elif t.startswith('global_load_lds') or (t.startswith('buffer_load') and ' lds' in t):
            ev.append('DMA')
        elif t.startswith(('global_load', 'buffer_load', 'flat_load')):
            ev.append('GL')
        elif t.startswith(('global_store', 'buffer_store', 'flat_store')):
            ev.append('GS')
        elif t.startswith('scratch_'):
            ev.append('SCR')
        elif t.startswith(('ds_read', 'ds_load')):
            ev.append('R')
        elif t.startswith(('ds_write', 'ds_store')):
            ev.append('W')
        elif t.startswith('ds_'):
            ev.append('DS')
        elif 'v_mfma' in t:
            ev.append('M')
        elif t.startswith('s_barrier'):
            ev.append('BAR')
        elif 'Loop Header' in l:
            ev.append('\nLOOP(' + l.split(':')[0].strip() + ')')
        elif t.startswith(('s_cbranch', 's_branch')):
            ev.append('br')
        elif t.startswith('v_'):
            ev.append('v')
    out = []
    for e in ev:
        if out and out[-1][0] == e:
            out[-1][1] += 1
        else:
            out.append([e, 1])
    return ' '.join(f'{e}x{n}' if n > 1 else e for e, n in out)


def main():
    fn, pat = sys.argv[1], sys.argv[2]
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
    lines = open(fn).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)] + [len(lines)]
    for a, b in zip(starts, starts[1:]):
        name = lines[a].split(':')[0]
        if pat not in name:
            continue
        body = lines[a:b]
        code = [l for l in body if 'codeLenInByte' in l]
        print('==', name, code[0].strip('; ') if code else '')
        print(events(body)[:limit])


if __name__ == '__main__':
    main()
