#!/bin/bash
# A/B of environment switches on the training-step benchmark, interleaved rounds in one gpurun call.
# usage: bash tools/gpu_ab.sh <tag> <rounds> "ENV1=a ENV2=b" "ENV1=c" ...     (each quoted string = one arm; "-" = no env)
tag=$1; rounds=$2; shift 2
out=gpurun_out/${tag}_ab.txt
: > $out
for r in $(seq 1 $rounds); do
  i=0
  for arm in "$@"; do
    envs=""; [ "$arm" != "-" ] && envs="$arm"
    ms=$(env $envs python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
    echo "round $r arm $i [$arm] $ms" | tee -a $out
    i=$((i+1))
  done
done
