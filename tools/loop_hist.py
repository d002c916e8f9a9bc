#!/usr/bin/env python3
"""Instruction histogram of the largest loop of a kernel in hipcc -S output.  usage: loop_hist.py file.s kernel_substring"""
import re
import sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
i0 = [i for i, l in enumerate(s) if sys.argv[2] in l and l.startswith('_Z') and l.split(';')[0].rstrip().endswith(':')][0]
end = next(i for i in range(i0, len(s)) if s[i].startswith('.Lfunc_end'))
body = s[i0:end]
labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'\.LBB\d+_\d+:', l)}
loops = []
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
print("kernel lines", len(body), "loops", loops)
a, b = max(loops, key=lambda t: t[1] - t[0])
c = Counter()
for l in body[a:b]:
    l = l.strip().split(';')[0].strip()
    if not l or l.startswith('.'):
        continue
    c[l.split()[0]] += 1
for k, v in c.most_common(60):
    print("%5d %s" % (v, k))
print("total", sum(c.values()))
