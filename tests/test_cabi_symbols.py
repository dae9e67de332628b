"""CPU-side check (no GPU, no compute): the C-ABI library builds for gfx950, loads, and exports every
symbol that include/amdnuwa.h declares; the ctypes signature table covers the same set."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as G
    from nuwa_pytorch_amd import build as B, _lib
    lib = B.build(verbose=False)
    so = ctypes.CDLL(lib)
    declared = G.declared_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(so, s)]
    assert not missing, missing
    assert set(_lib.SIGNATURES) == set(declared), set(_lib.SIGNATURES) ^ set(declared)
    assert so.amdnuwa_abi_version() == _lib.ABI_VERSION


def test_shipped_isa_has_no_packed_fp32_ops_and_no_new_spills():
    """ISA lint of the shipped code objects (tools/isa_lint.py): the library is compiled without packed fp32 VALU instructions -- the
    instruction class of the round-4 head-mix defect -- and the hot kernels stay inside their scratch bounds"""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import isa_lint
    from nuwa_pytorch_amd import build as B
    bad = isa_lint.check_shipped(B.build(verbose=False))
    assert not bad, bad


def test_argument_validation_without_gpu():
    """entry points reject bad descriptors before touching the device"""
    from nuwa_pytorch_amd import _lib
    L = _lib.lib()
    assert L.amdnuwa_gemm_nt(None, None) == -1
    d = _lib.GemmDesc()
    assert L.amdnuwa_gemm_nt(ctypes.byref(d), None) == -1            # null operands
    g = _lib.S3Geom()
    g.dim_head = 48
    assert L.amdnuwa_sparse3dna_bwd_workspace_bytes(ctypes.byref(g)) == 0
    assert L.amdnuwa_xattn_jp(256) == 288 and L.amdnuwa_xattn_jp(7) == 32
    assert b'workspace' in L.amdnuwa_error_string(-3)
    # the RCCL communicator entry points: argument checks come before librccl is touched
    h = ctypes.c_void_p()
    assert L.amdnuwa_comm_init(ctypes.byref(h), None, 128, 0, 1, 0) == -1
    assert L.amdnuwa_comm_init(ctypes.byref(h), ctypes.create_string_buffer(128), 128, 2, 2, 0) == -1     # rank >= world
    assert L.amdnuwa_comm_unique_id(ctypes.create_string_buffer(16), 16) == -1
    assert L.amdnuwa_comm_allreduce(None, None, 4, 1, None) == -1
    assert L.amdnuwa_comm_destroy(None) == 0
    assert b'RCCL' in L.amdnuwa_error_string(-4)


def test_no_product_import_of_oracle():
    """the product package must never import / reference the oracle"""
    pkg = os.path.join(ROOT, 'nuwa_pytorch_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src and 'nuwa_oracle' not in src, f


def test_integration_stub_names_the_current_abi_and_real_symbols():
    """INTEGRATION.md's reference-side binding must quote the ABI version the library reports and only call entry points it exports"""
    import re
    from nuwa_pytorch_amd import _lib
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r'amdnuwa_abi_version\(\) == (\d+)', text)
    assert m and int(m.group(1)) == _lib.ABI_VERSION
    called = set(re.findall(r'_L\.(amdnuwa_\w+)', text))
    assert called and called <= set(_lib.SIGNATURES) | {'amdnuwa_abi_version', 'amdnuwa_error_string'}, called - set(_lib.SIGNATURES)


def test_profiles_readme_lists_only_files_that_exist():
    """round-4 review: profiles/README.md listed seven files that were gone.  Every file named in the first column of its tables must exist."""
    import re
    pdir = os.path.join(ROOT, 'profiles')
    have = set(os.listdir(pdir))
    missing = []
    for line in open(os.path.join(pdir, 'README.md')):
        if not line.startswith('| `'):
            continue
        for name in re.findall(r'`([^`]+)`', line.split('|')[1]):
            if '*' not in name and name not in have:
                missing.append(name)
    assert not missing, missing


def test_committed_pmc_traffic_belongs_to_the_current_gemm_kernels():
    """bench.py refuses a PMC traffic figure taken on other kernel sources (roofline.traffic = null).  The figure of the headline
    configuration must therefore be re-taken (tools/gpu_final_r06.sh; copy gpurun_out/<TAG>_traffic.json over profiles/traffic.json)
    whenever csrc/gemm.hip changes: a stale file fails HERE, not silently in the round's final bench line."""
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, 'profiles', 'traffic.json')) as f:
        entry = json.load(f)['cfg3_b128_bf16x3-fwd']
    h = hashlib.sha256()
    with open(os.path.join(root, 'nuwa_pytorch_amd', 'csrc', 'gemm.hip'), 'rb') as f:
        h.update(f.read())
    assert entry['src_sha16'] == h.hexdigest()[:16], 'profiles/traffic.json is stale: re-take the FETCH_SIZE / WRITE_SIZE passes'
    assert entry['bytes_per_launch'] > 1e9
