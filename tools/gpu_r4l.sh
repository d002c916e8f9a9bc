#!/bin/bash
for k in 0 1 2 3 9; do
  ms=$(env B2S_LAB_SKIP_STREAMS=$k python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "plain skip $k: $ms"
done
