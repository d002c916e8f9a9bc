#!/usr/bin/env python3
"""Aggregate rocprofv3 --kernel-trace CSVs by (kernel name, grid size): launches, mean duration.
usage: trace_by_kernel.py a_kernel_trace.csv [b_kernel_trace.csv]  [--min-us X]"""
import csv
import re
import sys
from collections import OrderedDict

args = sys.argv[1:]
min_us = 0.0
if '--min-us' in args:
    i = args.index('--min-us')
    min_us = float(args[i + 1])
    del args[i:i + 2]
files = args


def short(name):
    name = re.sub(r'\(.*', '', name)
    name = re.sub(r'^void ', '', name)
    return name[-60:]


def load(path):
    agg = OrderedDict()
    for r in csv.DictReader(open(path)):
        key = (short(r['Kernel_Name']), int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1), int(r['Workgroup_Size_X']))
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    return agg


tabs = [load(f) for f in files]
keys = list(tabs[0].keys())
for t in tabs[1:]:
    for k in t:
        if k not in keys:
            keys.append(k)
tot = [0.0] * len(tabs)
for k in keys:
    cells = []
    for i, t in enumerate(tabs):
        c, us = t.get(k, (0, 0.0))
        tot[i] += us
        cells.append((c, us))
    if max(us for _, us in cells) < min_us:
        continue
    print("%-60s grid %8d wg %4d " % k + " ".join("n=%5d mean %8.1f tot %9.0f |" % (c, us / max(c, 1), us) for c, us in cells))
print("total kernel time (us): " + " ".join("%.0f" % t for t in tot))
