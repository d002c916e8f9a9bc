#!/bin/bash
# GPU_MAX_HW_QUEUES default / 4: plain step, forced data-parallel legs, c3 workload; interleaved rounds in one gpurun call (profiles/NOTES_r06.md section 8)
for r in 1 2; do
for q in default 4; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  a=$(python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  b=$(B2S_FORCE_DP=1 python bench.py --gpus 1 --no-cpu-baseline --no-roofline-pass --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], [l.get('ms_per_step') for l in d['dp_legs']])")
  c=$(python bench.py --workload c3 --no-cpu-baseline --no-roofline-pass --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "round $r GPU_MAX_HW_QUEUES=$q: step $a | forced DP $b | c3 $c"
done; done
