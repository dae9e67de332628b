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
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in counters:
        # MFMA-busy fraction per family = sum of SQ_VALU_MFMA_BUSY_CYCLES (cycles of matrix-pipe occupancy, summed over every SIMD of the chip: 16 per
        # v_mfma_f32_16x16x32) / (1024 SIMDs x kernel duration x 2.4 GHz), i.e. against the clock the 2.5 PFLOP/s peak is quoted at; durations
        # from the kernel trace of the SAME counter pass (pmc_kernel_trace.csv next to the counter CSV, joined on Dispatch_Id).
        import os
        busy, dur = defaultdict(float), defaultdict(float)
        nlaunch = defaultdict(int)
        for f in files:
            kt = os.path.join(os.path.dirname(f), 'pmc_kernel_trace.csv')
            rows = [r for r in csv.DictReader(open(f, newline='')) if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES']
            if not rows or not os.path.exists(kt):
                continue
            d = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9 for r in csv.DictReader(open(kt, newline=''))}
            seen = set()
            for r in rows:
                k = fam(r['Kernel_Name'])
                busy[k] += float(r['Counter_Value'])
                if r['Dispatch_Id'] not in seen and r['Dispatch_Id'] in d:
                    seen.add(r['Dispatch_Id']); dur[k] += d[r['Dispatch_Id']]; nlaunch[k] += 1
        if dur:
            print()
            print(f'{"kernel family":56s} {"launches":>8s} {"avg us (counter pass)":>22s} {"MFMA-busy fraction of the 2.4 GHz peak":>40s}')
            for k in sorted(dur, key=lambda x: -dur[x]):
                if busy[k] > 0:
                    print(f'{k:56s} {nlaunch[k]:8d} {dur[k] / nlaunch[k] * 1e6:22.1f} {busy[k] / (1024 * dur[k] * 2.4e9):40.3f}')
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
        # whole-step traffic: every kernel of the pass / the number of decoder steps in it (= launches of the once-per-step cross-entropy kernel)
        def allsum(c):
            return sum(v[c][1] for v in agg.values() if c in v)
        nsteps = max((v['FETCH_SIZE'][0] for k, v in agg.items() if k.startswith('ce_fwd') and 'FETCH_SIZE' in v), default=0)
        step_bytes = ((2.0 * allsum('FETCH_SIZE') + allsum('WRITE_SIZE')) * 1024.0 / nsteps) if nsteps else None
        cur[key] = {'bytes_per_launch': (2.0 * fs + ws) * 1024.0, 'fetch_kib_raw': fs, 'write_kib_raw': ws, 'step_bytes': step_bytes, 'steps_in_pass': nsteps,
                    'src_sha16': bench.file_sha16(bench.ROOFLINE_SOURCES),
                    'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), avg over gemm_nt_* launches; read bytes = 2 x FETCH_SIZE (gfx950 correction), write bytes = WRITE_SIZE'}
        json.dump(cur, open(out, 'w'), indent=1)
        print('wrote', out, key, cur[key]['bytes_per_launch'])


if __name__ == '__main__':
    main()
