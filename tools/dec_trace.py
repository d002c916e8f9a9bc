#!/usr/bin/env python3
"""Gaps between the kernels of the decode frame loop from a rocprofv3 kernel trace: usage dec_trace.py t_kernel_trace.csv"""
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
df = [r for r in rows if 'k_df_' in r['Kernel_Name'] or 'k_dec_' in r['Kernel_Name'] or 'skinny' in r['Kernel_Name']]
# one frame = from k_df_prenet to the next
idx = [i for i, r in enumerate(df) if 'prenet' in r['Kernel_Name']]
if len(idx) < 10: sys.exit("no frames")
mid = idx[len(idx) * 3 // 4]; nxt = idx[len(idx) * 3 // 4 + 1]
fr = df[mid:nxt + 1]
t0 = int(fr[0]['Start_Timestamp']); prev_end = None
for r in fr:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void (anonymous namespace)::', '')
    print("%8.1f  %-50s dur %6.1f  gap %5.1f" % ((s - t0) / 1e3, name[:50], (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3))
    prev_end = e
span = [(int(df[idx[i + 1]]['Start_Timestamp']) - int(df[idx[i]]['Start_Timestamp'])) / 1e3 for i in range(len(idx) - 1)]
span.sort()
print("frames %d: frame period us: median %.1f  p10 %.1f  p90 %.1f" % (len(span), span[len(span) // 2], span[len(span) // 10], span[len(span) * 9 // 10]))
