import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
df = [r for r in rows if 'k_df_' in r['Kernel_Name']]
# sequence per layer: self, cross, ffn, ffn(2nd), cross(2nd: weights hot? no: after ffn), cross(3rd)
seq = {}
prev = None; run = 0
for r in df[len(df)//2:]:
    n = re.sub(r'\(.*', '', r['Kernel_Name']).split('::')[-1]
    run = run + 1 if n == prev else 0
    prev = n
    seq.setdefault((n, run), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(seq.items()):
    v.sort(); print(k, "n=%d median %.1f us" % (len(v), v[len(v)//2]))
