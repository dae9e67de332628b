#!/usr/bin/env python
"""Per-kernel-family averages of rocprofv3 --pmc counters (one CSV per pass, `--output-format csv`).
    python tools/pmc_summary.py gpurun_out/pmcX/*_counter_collection.csv [--json profiles/traffic.json --key cfg3_b32]
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950
tallies a wide (16 B/lane) coalesced read at HALF its bytes, so the corrected read traffic is 2 x FETCH_SIZE; WRITE_SIZE
is used as reported (uncalibrated).  The --json mode writes the NT-GEMM family's bytes per launch for bench.py."""
import csv
import json
import re
import sys
from collections import defaultdict


def fam(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'[<(].*$', '', n)
    return n[:70]


def main():
    files = [a for a in sys.argv[1:] if a.endswith('.csv')]
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in files:
        with open(f, newline='') as fh:
            for r in csv.DictReader(fh):
                a = agg[fam(r['Kernel_Name'])][r['Counter_Name']]
                a[0] += 1
                a[1] += float(r['Counter_Value'])
    counters = sorted({c for k in agg.values() for c in k})
    print(f'{"kernel family":56s} {"launches":>8s} ' + ' '.join(f'{c:>22s}' for c in counters))
    keys = sorted(agg, key=lambda k: -sum(v[1] for v in agg[k].values()))
    for k in keys:
        n = max(v[0] for v in agg[k].values())
        if not (k.startswith(('gemm_', 's3_', 'xattn', 'ln_', 'geglu', 'conv2d', 'vq_', 'groupnorm', 'embed', 'ce_', 'partial', 'splitk', 'cast', 'transpose', 'colsum'))):
            continue
        print(f'{k:56s} {n:8d} ' + ' '.join(f'{(agg[k][c][1] / agg[k][c][0]) if c in agg[k] else float("nan"):22.1f}' for c in counters))
    if all(c in counters for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES')):
        # MFMA-busy fraction per family.  Normalisation (checked against the round-2 passes: NT GEMMs 0.26 next to 0.26 of the MFMA peak in
        # algorithmic FLOP/s, cross-attention backward 0.12): SQ_VALU_MFMA_BUSY_CYCLES is summed over the 256 CUs, SQ_BUSY_CYCLES over 64 SQ
        # instances, so busy fraction of one CU's matrix pipe = MFMA_BUSY / 256 / (SQ_BUSY / 64) = MFMA_BUSY / (4 SQ_BUSY).
        print()
        print(f'{"kernel family":56s} {"launches":>8s} {"MFMA busy fraction":>20s} {"HBM GB per launch (2 FETCH + WRITE)":>36s}')
        for k in keys:
            a = agg[k]
            if 'SQ_VALU_MFMA_BUSY_CYCLES' not in a or 'SQ_BUSY_CYCLES' not in a or not k.startswith(('gemm_', 's3_', 'xattn', 'ln_', 'ce_', 'splitk', 'embed')):
                continue
            mf, sb = a['SQ_VALU_MFMA_BUSY_CYCLES'][1] / a['SQ_VALU_MFMA_BUSY_CYCLES'][0], a['SQ_BUSY_CYCLES'][1] / a['SQ_BUSY_CYCLES'][0]
            gb = ''
            if 'FETCH_SIZE' in a and 'WRITE_SIZE' in a:
                gb = f"{(2 * a['FETCH_SIZE'][1] / a['FETCH_SIZE'][0] + a['WRITE_SIZE'][1] / a['WRITE_SIZE'][0]) * 1024 / 1e9:.3f}"
            print(f'{k:56s} {max(v[0] for v in a.values()):8d} {mf / (4 * sb) if sb else float("nan"):20.3f} {gb:>36s}')
    if '--json' in sys.argv:
        out, key = sys.argv[sys.argv.index('--json') + 1], sys.argv[sys.argv.index('--key') + 1]
        nt = [k for k in agg if k.startswith('gemm_nt_')]
        def tot(c):
            n = sum(agg[k][c][0] for k in nt if c in agg[k])
            return (sum(agg[k][c][1] for k in nt if c in agg[k]) / n) if n else None
        fs, ws = tot('FETCH_SIZE'), tot('WRITE_SIZE')
        if fs is None or ws is None:
            raise SystemExit('need FETCH_SIZE and WRITE_SIZE passes')
        try:
            cur = json.load(open(out))
        except (OSError, ValueError):
            cur = {}
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench                          # the hash bench.py will compare against: a figure from other kernel sources is refused
        cur[key] = {'bytes_per_launch': (2.0 * fs + ws) * 1024.0, 'fetch_kib_raw': fs, 'write_kib_raw': ws,
                    'src_sha16': bench.file_sha16(bench.ROOFLINE_SOURCES),
                    'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), avg over gemm_nt_* launches; read bytes = 2 x FETCH_SIZE (gfx950 correction), write bytes = WRITE_SIZE'}
        json.dump(cur, open(out, 'w'), indent=1)
        print('wrote', out, key, cur[key]['bytes_per_launch'])


if __name__ == '__main__':
    main()
