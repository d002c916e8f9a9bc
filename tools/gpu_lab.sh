#!/bin/bash
# One gpurun call = an optional pytest selection + an interleaved A/B of environment switches on the training-step benchmark.
# usage (through gpurun): bash tools/gpu_lab.sh <tag> "<pytest arguments or ->" <rounds> "ENV=a" "ENV=b ENV2=c" "-" ...
#   e.g. gpurun -- 'bash tools/gpu_lab.sh tail "tests/test_gpu_trainer_state.py -q" 3 "B2S_TAIL_ADAM=0" "-"'
# (A/B arms always inside ONE call: boxes differ by 2-3 %.  Traces: tools/gpu_trace.sh; counters: tools/gpu_pmc.sh; the round's
# profile set: tools/gpu_round_profiles.sh.)
tag=$1; sel=$2; rounds=$3; shift 3
mkdir -p gpurun_out
if [ "$sel" != "-" ]; then python -m pytest $sel 2>&1 | tail -15 > gpurun_out/${tag}_tests.log; tail -3 gpurun_out/${tag}_tests.log; fi
[ "$rounds" -gt 0 ] && bash tools/gpu_ab.sh $tag $rounds "$@"
