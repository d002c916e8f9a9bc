#!/usr/bin/env python3
"""Print the memory / MFMA / wait skeleton of the innermost loop of a kernel from hipcc -S output.
usage: loop_isa.py file.s kernel_substring [all]"""
import re
import sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
i0 = [i for i, l in enumerate(s) if key in l and l.startswith('_Z') and l.split(';')[0].rstrip().endswith(':')][0]
end = next(i for i in range(i0, len(s)) if 's_endpgm' in s[i])
body = s[i0:end]
hdr = [i for i, l in enumerate(body) if 'Inner Loop Header' in l]
print("kernel at line %d, %d lines, loop headers at %s" % (i0, len(body), hdr[:6]))
pat = r'(s_waitcnt|s_barrier|ds_read|ds_write|global_load|global_store|buffer_|v_mfma|s_cbranch|scratch_|s_setprio)'
for h in hdr[:int(sys.argv[4]) if len(sys.argv) > 4 else 1]:
    n_valu = 0
    for i in range(h, len(body)):
        l = body[i].strip().split(';')[0].strip()
        if not l:
            continue
        if re.match(pat, l) or len(sys.argv) > 3 and sys.argv[3] == 'all':
            if n_valu:
                print("      ... %d other instr" % n_valu)
                n_valu = 0
            print(i, l[:70])
        elif not l.startswith('.'):
            n_valu += 1
        if l.startswith('s_cbranch') and i > h + 20:
            break
