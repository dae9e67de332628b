#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE of tools/pmc_calib.hip (every kernel moves exactly 2 GiB per launch) -> counter bytes / true bytes per
access pattern.  rocprofv3 reports both counters in KiB.
    python tools/pmc_calib_summary.py <fetch pass csv> <write pass csv>"""
import csv
import re
import sys
from collections import defaultdict

TRUE = float(2 << 30)
agg = defaultdict(lambda: defaultdict(list))
for f in sys.argv[1:]:
    with open(f, newline='') as fh:
        for r in csv.DictReader(fh):
            k = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '')
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']) * 1024.0)
print(f'{"pattern":28s} {"FETCH_SIZE / true":>18s} {"WRITE_SIZE / true":>18s}   (true = 2 GiB per launch; median of the launches)')
for k in sorted(agg):
    med = lambda c: sorted(agg[k][c])[len(agg[k][c]) // 2] / TRUE if agg[k].get(c) else float('nan')
    print(f'{k:28s} {med("FETCH_SIZE"):18.3f} {med("WRITE_SIZE"):18.3f}')
