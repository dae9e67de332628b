#!/usr/bin/env python
"""Per-kernel summary (calls, total / avg / min / max duration, % of GPU time) from a rocprofv3 rocpd
SQLite database (`rocprofv3 --kernel-trace --stats` on ROCm 7.2 writes *_results.db).
    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [--last-frac 0.6] > profiles/xxx.txt"""
import re
import sqlite3
import sys


_DM = {}


def demangle(n):
    """rocpd stores mangled symbols with a .kd suffix; c++filt them (cached)"""
    if n not in _DM:
        m = n[:-3] if n.endswith('.kd') else n
        try:
            import subprocess
            _DM[n] = subprocess.run(['c++filt', m], capture_output=True, text=True, timeout=10).stdout.strip() or m
        except Exception:
            _DM[n] = m
    return _DM[n]


def short(n):
    n = demangle(n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*$', '', n)
    return n[:90]


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute('select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d '
                       'join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start').fetchall()
    if not rows:
        print('no kernel dispatches'); return
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    frac = 0.0
    if '--last-frac' in sys.argv:
        frac = 1.0 - float(sys.argv[sys.argv.index('--last-frac') + 1])
    cut = t0 + (t1 - t0) * frac
    rows = [r for r in rows if r[1] >= cut]
    agg = {}
    for name, s, e, gx, wx in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    span = max(r[2] for r in rows) - rows[0][1]
    print(f'# {path}: {len(rows)} dispatches, GPU busy {tot / 1e6:.3f} ms over a {span / 1e6:.3f} ms span ({100.0 * tot / span:.1f}% busy)')
    print(f'{"kernel":90s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"min_us":>9s} {"max_us":>9s} {"pct":>6s}')
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{k:90s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100.0 * a[1] / tot:6.2f}')
    # the same, template instantiations folded into their kernel family (gemm_nt_256_kernel<...> -> gemm_nt_256_kernel)
    fam = {}
    for k, a in agg.items():
        f = fam.setdefault(re.sub(r'<.*$', '', k), [0, 0, 1 << 62, 0])
        f[0] += a[0]; f[1] += a[1]; f[2] = min(f[2], a[2]); f[3] = max(f[3], a[3])
    print()
    print('# by kernel family (template arguments folded); gemm_nt_* together = the bench.py roofline kernel')
    print(f'{"family":60s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"pct":>6s}')
    for k, a in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f'{k:60s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.2f} {100.0 * a[1] / tot:6.2f}')
    nt = [a for k, a in fam.items() if k.startswith('gemm_nt_')]
    if nt:
        c, t = sum(a[0] for a in nt), sum(a[1] for a in nt)
        print(f'{"gemm_nt_* (all NT GEMM launches)":60s} {c:7d} {t / 1e6:10.3f} {t / c / 1e3:9.2f} {100.0 * t / tot:6.2f}')


if __name__ == '__main__':
    main()
