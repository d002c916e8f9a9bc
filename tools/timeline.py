#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: kernels in start order with duration, the gap to the
previous kernel's end on the same queue, and a per-category summary.  A step runs from one k_embed_prep_fwd launch to the next.
usage: timeline.py t_kernel_trace.csv [--step N] [--list]"""
import csv
import re
import sys
from collections import OrderedDict

args = sys.argv[1:]
lst = '--list' in args
if lst:
    args.remove('--list')
step = -2
if '--step' in args:
    i = args.index('--step'); step = int(args[i + 1]); del args[i:i + 2]
rows = list(csv.DictReader(open(args[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    m = re.match(r'([\w:]+(<[^(]*>)?)', n)
    return (m.group(1) if m else n)[:64]


# a step starts at the encoder's embedding kernel (the first kernel of the forward pass)
adam = [i for i, r in enumerate(rows) if 'k_embed_prep_fwd' in r['Kernel_Name']]
lo, hi = adam[step - 1], adam[step]
seg = rows[lo:hi]
t0 = int(seg[0]['Start_Timestamp'])
t_end = max(int(r['End_Timestamp']) for r in seg)
print("step: %d kernels, %.3f ms from first start to last end" % (len(seg), (t_end - t0) / 1e6))
last_end = {}
cat = OrderedDict()
gap_tot = {}
for r in seg:
    q = r['Queue_Id']
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = max(e, last_end.get(q, 0))
    name = short(r['Kernel_Name'])
    grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))
    if lst:
        print("%9.1f q%s %-64s wg %5d  dur %7.1f  gap %6.1f" % ((s - t0) / 1e3, q, name, grid, (e - s) / 1e3, gap))
    key = re.sub(r'<.*', '', name)
    if 'gemm' in key:
        key = name
    c = cat.setdefault(key, [0, 0.0])
    c[0] += 1; c[1] += (e - s) / 1e3
    gap_tot[q] = gap_tot.get(q, 0.0) + max(gap, 0.0)
tot = 0.0
for k, (n, us) in sorted(cat.items(), key=lambda kv: -kv[1][1]):
    print("%-66s n=%4d  %8.1f us  (%6.1f avg)" % (k, n, us, us / n))
    tot += us
print("sum of kernel durations: %.3f ms; gaps per queue (us): %s" % (tot / 1e3, {q: round(v, 1) for q, v in gap_tot.items()}))
