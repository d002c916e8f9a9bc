#!/bin/bash
# step time vs per-GPU batch size (through gpurun): bash tools/batch_sweep.sh 7 14 28 56
for b in "$@"; do
  python bench.py --batch $b --no-cpu-baseline --no-roofline-pass --no-extras --steps 20 --warmup 5 2>/dev/null | B=$b python -c "
import sys, json, os
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('B=%s ms_per_step=%.3f frac_of_bf16_peak=%.4f' % (os.environ['B'], d['ms_per_step'], d['roofline_step']['frac']))"
done
