#!/bin/bash
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 B2S_FORCE_DP=1
p=29700
for k in 0 1 2 3 4 5 6 7; do
  p=$((p+1))
  ms=$(env MASTER_PORT=$p B2S_LAB_SKIP_STREAMS=$k python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "skip $k: $ms"
done
