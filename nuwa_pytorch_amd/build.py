"""Builds libamdnuwa.so (gfx950) in-tree with hipcc.  No torch involvement: the library is a plain
C-ABI shared object (include/amdnuwa.h) loaded through ctypes by nuwa_pytorch_amd._lib.

    python -m nuwa_pytorch_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libamdnuwa.so')
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'amdnuwa.h')
SOURCES = ['api.hip', 'gemm.hip', 'elementwise.hip', 'sparse3dna.hip', 'xattn.hip', 'xattn2.hip', 'xattn6.hip', 'vae.hip', 'optim.hip', 'decode.hip', 'comm.hip']
# One generic `gfx950` code object (loads whatever the device's XNACK mode).  Measured at the end of round 6 (profiles/r06zn_xnack_ab.txt): a code object built for
# XNACK-off devices only (AMDNUWA_BUILD_ARCH=gfx950:xnack-) runs the step 0.5 % faster in an A/B/A/B of one call (457.0 -> 454.4 ms: the generic object must stay
# safe under XNACK replay, e.g. a load's destination may not overlap its address registers) -- an opt-in for deployments that never enable demand paging; it
# does not load on a device running with HSA_XNACK=1, and a library that also carried the xnack+ object is refused by this pool's job runner.
# AMDNUWA_BUILD_ARCH: comma-separated --offload-arch list (A/B runs and that opt-in).
ARCHS = os.environ.get('AMDNUWA_BUILD_ARCH', 'gfx950').split(',')
# Flags of every device compile of the SHIPPED library: no packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).
# Round 4 found the head-mix loop of the two-row Sparse3DNA forward tile returning wrong LOW halves out of a packed-FMA sequence once two
# workgroups shared a CU.  Round 5 separated the two cures on the GPU (profiles/r05a_mix_variants.txt): packed ops ON + loops written freely
# -> 431 k differing elements in 8 runs; packed ops OFF + the same free loops -> bit-identical; so the instruction class is the necessary
# ingredient, and the whole library is compiled without it (+1 % on the step: two kernels pay).  tools/isa_lint.py --forbid-pk
# (run by __graft_entry__.build() and tests/test_cabi_symbols.py) fails if a packed op reappears.
DEFAULT_FLAGS = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def DEFAULT_FLAGS_FOR(variant):
    """flags of the shipped build that a variant may switch off ('pk' = the shipped build WITHOUT DEFAULT_FLAGS, for A/B runs)"""
    return [] if variant in ('pk', 'pk_nofix') else DEFAULT_FLAGS


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


# per-file compiler options.  xattn2.hip: its one-wave-per-SIMD kernels hold > 256 live registers; by default the compiler puts EVERY
# MFMA result into AGPRs there and moves the transient ones (scores, head mixes: consumed by the VALU right away) back with
# v_accvgpr_read, zero-initialises their accumulators with v_accvgpr_write, and parks MFMA-only operands in AGPRs to read them back
# before each use: 1300 v_accvgpr moves in xattn3_bwd.  With the VGPR form the transient results land where they are used (290 moves,
# 15 % fewer instructions in a kernel that is bound by instruction issue); the long-lived accumulators still sit in AGPRs.
EXTRA_FLAGS = {'xattn2.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form'], 'xattn6.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}

# Build variants (A/B runs of compiler options: `python -m nuwa_pytorch_amd.build --variant pk` writes lib_pk/libamdnuwa.so, which
# AMDNUWA_LIBRARY=... then selects).  'pk' / 'pk_nofix': WITH packed fp32 ops (DEFAULT_FLAGS dropped), the Sparse3DNA head-mix loops pinned
# (round 4's cure) / written freely (fails the stress: the experiment that separated the two cures); 'pin': the shipped flags + the pin.
# The host pass ignores the feature flag.
NOPK_FLAGS = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
VARIANTS = {'': [], 'pk': ['-DS3_MIX_PIN=1'], 'pk_nofix': ['-DS3_MIX_PIN=0'], 'pin': ['-DS3_MIX_PIN=1'], 'x6t': ['-DX6_TIMING=1'],
            # round 6, call zm: the backend's other scheduling strategies over the whole library (the hand-placed loops are pinned with sched_barrier and do not move)
            'ilp': ['-mllvm', '-amdgpu-sched-strategy=max-ilp'], 'mclause': ['-mllvm', '-amdgpu-sched-strategy=max-memory-clause'],
            'xoff': []}        # (with AMDNUWA_BUILD_ARCH=gfx950:xnack- : a code object for XNACK-off devices only, see ARCHS)


def lib_path(variant=''):
    return os.path.join(HERE, 'lib' + ('_' + variant if variant else ''), 'libamdnuwa.so')


def build(force=False, verbose=True, variant=''):
    LIBDIR = os.path.dirname(lib_path(variant))
    LIB = lib_path(variant)
    vflags = VARIANTS[variant] if variant in VARIANTS else variant.split()
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    common = [os.path.join(CSRC, 'common.h'), HEADER]
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src] + common + [os.path.abspath(__file__)]):
            jobs.append([_hipcc()] + [f'--offload-arch={a}' for a in ARCHS] + ['-O3', '-std=c++17', '-fPIC'] + EXTRA_FLAGS.get(s, []) + DEFAULT_FLAGS_FOR(variant) + vflags + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed:\n{" ".join(cmd)}\n{r.stdout}\n{r.stderr}')

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([_hipcc()] + [f'--offload-arch={a}' for a in ARCHS] + ['-shared', '-fPIC', '-o', LIB] + objs + ['-ldl'])
    return LIB


if __name__ == '__main__':
    var = sys.argv[sys.argv.index('--variant') + 1] if '--variant' in sys.argv else ''
    print(build(force='--force' in sys.argv, variant=var))
