#!/usr/bin/env python3
"""Aggregate B2S_PROF_DUMP files (one line per GEMM launch: variant M N K batch splitk us TF) by shape.
usage: prof_by_shape.py a.txt [b.txt]   -- with two files prints them side by side (same launch sequence expected)"""
import sys
from collections import OrderedDict


def load(path):
    agg = OrderedDict()
    for line in open(path):
        f = line.split()
        key = tuple(int(x) for x in f[:6])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += float(f[6])
    return agg


tabs = [load(p) for p in sys.argv[1:]]
keys = list(tabs[0].keys())
tot = [0.0] * len(tabs)
print("var      M     N     K  b sk    n " + "".join("%10s" % ("us[%d]" % i) for i in range(len(tabs))) + "   total us per file")
for k in keys:
    row = "%3d %6d %5d %5d %2d %2d" % k
    n = tabs[0][k][0]
    row += " %4d " % n
    for i, t in enumerate(tabs):
        c, us = t.get(k, (0, 0.0))
        row += "%10.1f" % (us / max(c, 1))
        tot[i] += us
    row += "   " + " ".join("%9.0f" % t.get(k, (0, 0.0))[1] for t in tabs)
    print(row)
print("sum of launch durations (us): " + " ".join("%.0f" % t for t in tot))
