#!/usr/bin/env python
"""ISA lint of the SHIPPED device code: unbundle libamdnuwa.so's gfx950 code objects (into a scratch directory, never next to the
library), disassemble them and check, per kernel,

  * packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  Round 4 found the head-mix loop of the two-row
    Sparse3DNA forward tile storing wrong LOW halves out of such a sequence once two workgroups shared a CU; the mechanism was never
    isolated (a micro-kernel of the suspected shape ran 4.2 G trials clean), so round 5 takes the whole instruction class out of the
    library: it is built with the packed-fp32 target feature off (nuwa_pytorch_amd/build.py DEFAULT_FLAGS) and `--forbid-pk` makes
    any reappearance -- a compiler bump, a dropped flag, a new file compiled without it -- a failure.
  * the narrower shape the round-4 hypothesis named: a packed op whose destination pair is also a source pair read ACROSS halves
    (op_sel / op_sel_hi); reported whenever packed ops exist (`--forbid-overlap`).
  * scratch (spill) bytes per lane of every kernel, from the kernel descriptors' metadata (`--max-scratch N` fails above N).

    python tools/isa_lint.py [path/to/libamdnuwa.so] [--forbid-pk] [--forbid-overlap] [--max-scratch BYTES] [--top 12]
    python tools/isa_lint.py --m0 file.s ...      (hipcc -S listings: the M0 discipline around the inline-asm LDS-DMA pieces, see m0_discipline)

Used by tests/test_cabi_symbols.py::test_shipped_isa_has_no_packed_fp32_ops (CPU, no GPU needed) and by __graft_entry__.build()."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
PK_OPS = ('v_pk_fma_f32', 'v_pk_mul_f32', 'v_pk_add_f32')


def _pair(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    return (int(m.group(1)), int(m.group(2))) if m else None


def _sel(t, name, n, default):
    m = re.search(name + r':\[([01,]+)\]', t)
    if not m:
        return [default] * n
    v = [int(x) for x in m.group(1).split(',')]
    return v + [default] * (n - len(v))


def overlap(line):
    """True when a packed fp32 op overwrites a source pair it reads across halves"""
    t = line.strip()
    op = t.split()[0]
    body = re.split(r'\s+(?:op_sel|neg_lo|neg_hi|clamp)', t[len(op):])[0]
    body = body.split('//')[0]
    ops = [x.strip() for x in body.split(',')]
    dst, srcs = _pair(ops[0]), ops[1:]
    lo, hi = _sel(t, 'op_sel', len(srcs), 0), _sel(t, 'op_sel_hi', len(srcs), 1)
    return any(dst and _pair(s) == dst and (lo[k] == 1 or hi[k] == 0) for k, s in enumerate(srcs))


def code_objects(lib, scratch):
    """copies `lib` into `scratch`, unbundles it there, returns the gfx950 code-object paths"""
    dst = os.path.join(scratch, os.path.basename(lib))
    shutil.copy(lib, dst)
    subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', dst], capture_output=True, cwd=scratch, check=False)
    # (also the code objects of a build with target features in its name, e.g. gfx950:xnack-: AMDNUWA_BUILD_ARCH)
    return sorted(glob.glob(dst + '.*gfx950*'))


def scan(lib):
    """-> {kernel: dict(pk=int, overlap=int, example=str|None)}"""
    out = {}
    scratch = tempfile.mkdtemp(prefix='amdnuwa_isa_')
    try:
        cos = code_objects(lib, scratch)
        if not cos:
            raise RuntimeError(f'no gfx950 code object found in {lib}')
        for co in cos:
            dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--mcpu=gfx950', co], capture_output=True, text=True, check=True).stdout
            kernel = None
            for raw in dis.splitlines():
                m = re.match(r'^[0-9a-f]+ <([^>]+)>:', raw)
                if m:
                    kernel = m.group(1)
                    out.setdefault(kernel, dict(pk=0, overlap=0, example=None))
                    continue
                t = raw.strip()
                if not t or kernel is None:
                    continue
                op = t.split()[0]
                if op in PK_OPS:
                    e = out[kernel]
                    e['pk'] += 1
                    if overlap(t):
                        e['overlap'] += 1
                        e['example'] = e['example'] or t.split('//')[0].strip()
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    return out


def kernel_scratch(lib):
    """{kernel: scratch bytes per lane}: parsed per kernel entry of the metadata (order-independent)"""
    res = {}
    scratch = tempfile.mkdtemp(prefix='amdnuwa_isa_')
    try:
        for co in code_objects(lib, scratch):
            meta = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True, check=False).stdout
            # kernel entries start at "  - .agpr_count" / "  - .args" style list items: split on list-item starts at that indent
            for ent in re.split(r'\n\s{2,4}- (?=\.)', meta):
                n = re.search(r'\.name:\s+(\S+)', ent)
                p = re.search(r'\.private_segment_fixed_size:\s+(\d+)', ent)
                if n and p and '.kernarg_segment_size' in ent:
                    res[n.group(1)] = int(p.group(1))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    return res


# scratch (spill) bounds of the hot kernels of the training step, bytes per lane: a change that pushes one of them over its register budget
# shows up here (CPU, seconds) instead of as a slower step on the GPU.  Round 5 lesson: two run-time branches added to the bf16 epilogue
# cost gemm_nt_256p_kernel<1, true> 616 B of scratch per lane and the step 17 %, with every test green.
SCRATCH_BOUNDS = [('gemm_nt_256p_kernel', 0), ('gemm_nt_256_kernel', 0), ('gemm_nt_w4k_kernel', 16), ('gemm_tn_w4k_kernel', 0), ('gemm_tn_256_kernel', 0),
                  ('gemm_nt_256x3_kernel', 0), ('ln_post_pre_kernel', 0), ('ln_bwd_chain_kernel', 0), ('ln_fwd_kernel', 0), ('s3_fwd_tile_kernel', 0),
                  ('s3_fwd_mfma_kernel', 0), ('s3_bwd_q_mfma_kernel', 60), ('s3_bwd_kv_mfma_kernel', 0), ('xattn4_fwd_kernel', 0), ('xattn3_bwd_kernel', 272),
                  ('ce_fwd_reg_kernel', 0), ('splitk_reduce_kernel', 0)]


def check_shipped(lib=None):
    """the build / test gate: list of failures (empty = fine) -- packed fp32 VALU ops anywhere, or a hot kernel above its scratch bound"""
    lib = lib or os.path.join(ROOT, 'nuwa_pytorch_amd', 'lib', 'libamdnuwa.so')
    bad = []
    res = scan(lib)
    npk = sum(e['pk'] for e in res.values())
    if npk:
        worst = max(res, key=lambda k: res[k]['pk'])
        bad.append(f'{npk} packed fp32 VALU ops in {sum(1 for e in res.values() if e["pk"])} kernels (e.g. {res[worst]["pk"]} in {worst[:80]})')
    sc = kernel_scratch(lib)
    for k, v in sc.items():
        for name, bound in SCRATCH_BOUNDS:
            if name in k and v > bound:
                bad.append(f'{v} B/lane of scratch in {k[:100]} (bound {bound})')
    return bad


M0_READERS = re.compile(r'^(global_load_lds_|buffer_load_\w+.*\blds\b|s_movrel|v_movrel|ds_gws|s_sendmsg|v_interp)')


def m0_discipline(asm_text):
    """`hipcc -S` text -> list of violations of the rule that makes the `"m0"` clobber of dma16_asm (csrc/common.h) harmless: the compiler
    never carries an M0 value of its own ACROSS an inline-asm block that writes M0.  Checked per function in linear order:
      * an inline-asm reader of M0 (LDS-DMA piece) must follow an M0 write inside the same asm block;
      * a compiler-generated reader (LDS-DMA builtin, buffer_load ... lds, movrel, sendmsg, gws) must follow a compiler-generated M0 write
        with no asm M0 write in between;
      * a compiler-generated reader that takes its M0 from ANOTHER basic block (waterfall loops do) is only accepted in functions that
        contain no asm M0 write at all (a back edge could otherwise bring a stale value around)."""
    bad, func, src, in_asm, asm_wrote, block_has_write = [], None, None, False, False, False
    cross, asm_funcs = [], set()
    for n, raw in enumerate(asm_text.splitlines(), 1):
        t = raw.strip()
        if not t:
            continue
        if re.match(r'^[A-Za-z_.$][\w.$]*:', t):                  # a label: function entry or basic-block start
            if not t.startswith('.L'):
                func, src = t.split(':')[0], None
            block_has_write = False
            continue
        if t.startswith(';'):
            if 'ASMSTART' in t:
                in_asm, asm_wrote = True, False
            elif 'ASMEND' in t:
                in_asm = False
                if asm_wrote:
                    src = 'asm-stale'                             # whatever the compiler had in M0 is gone
            continue
        if t.startswith('.'):
            continue
        op = t.split()[0]
        if re.match(r'^(s_cbranch|s_branch|s_setpc|s_swappc|s_endpgm)', op):
            block_has_write = False
            continue
        writes = bool(re.match(r'^\S+\s+m0\b', t)) or op == 's_set_gpr_idx_on'
        if M0_READERS.match(t):
            if in_asm:
                if src != 'asm':
                    bad.append((func, n, t, src))
            else:
                if src != 'compiler':
                    bad.append((func, n, t, src))
                elif not block_has_write:
                    cross.append((func, n, t, 'compiler, another block'))
        if writes:
            src = 'asm' if in_asm else 'compiler'
            block_has_write = True
            if in_asm:
                asm_wrote = True
                asm_funcs.add(func)
    return bad + [c for c in cross if c[0] in asm_funcs]


def main(argv):
    if argv and argv[0] == '--m0':
        rc = 0
        for path in argv[1:]:
            bad = m0_discipline(open(path).read())
            print(f'{path}: {len(bad)} M0 readers without an M0 write of their own party in their basic block')
            for f, n, t, src in bad[:10]:
                print(f'   {f} line {n}: {t}   (M0 from: {src})')
            rc |= bool(bad)
        return rc
    valued = {'--top': 12, '--max-scratch': None}
    args, i = [], 0
    while i < len(argv):
        if argv[i] in valued:
            valued[argv[i]] = int(argv[i + 1]); i += 2
            continue
        if not argv[i].startswith('--'):
            args.append(argv[i])
        i += 1
    lib = args[0] if args else os.path.join(ROOT, 'nuwa_pytorch_amd', 'lib', 'libamdnuwa.so')
    top, max_scratch = valued['--top'], valued['--max-scratch']
    res = scan(lib)
    sc = kernel_scratch(lib)
    npk = sum(e['pk'] for e in res.values())
    nov = sum(e['overlap'] for e in res.values())
    print(f'{lib}: {len(res)} kernels, {npk} packed fp32 VALU ops in {sum(1 for e in res.values() if e["pk"])} kernels, '
          f'{nov} of them overwrite a source pair read across halves ({sum(1 for e in res.values() if e["overlap"])} kernels)')
    for k in sorted(res, key=lambda x: -res[x]['pk'])[:top]:
        e = res[k]
        if e['pk']:
            print(f'  {e["pk"]:6d} packed, {e["overlap"]:5d} overlapping  {k[:110]}' + (f'\n           e.g. {e["example"]}' if e['example'] else ''))
    spill = {k: v for k, v in sc.items() if v}
    print(f'kernels with scratch: {len(spill)} of {len(sc)}')
    for k in sorted(spill, key=lambda x: -spill[x])[:top]:
        print(f'  {spill[k]:6d} B/lane  {k[:110]}')
    rc = 0
    if '--forbid-pk' in argv and npk:
        print('FAIL: packed fp32 VALU ops present'); rc = 1
    if '--forbid-overlap' in argv and nov:
        print('FAIL: packed fp32 ops that overwrite a source pair read across halves'); rc = 1
    if max_scratch is not None and spill and max(spill.values()) > max_scratch:
        print(f'FAIL: scratch above {max_scratch} B/lane'); rc = 1
    return rc


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
