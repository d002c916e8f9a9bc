#!/bin/bash
# HBM traffic of the decode frame loop from the L2's memory-side counters: two separate rocprofv3 --pmc passes over the 64 x 1000-frame
# job (never combined with trace domains other than the kernel trace).  Usage (through gpurun): bash tools/gpu_pmc_decode.sh <tag>
tag=${1:-pmcd}
repo=$PWD
out=$repo/gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_$c -o p -- python $repo/bench.py --mode decode --steps 10 --warmup 1 --no-cpu-baseline > $out/${tag}_$c.log 2>&1
  rm -f $out/${tag}_$c/*kernel_trace.csv
done
cd $repo
python tools/pmc_traffic.py $out/${tag}_FETCH_SIZE/p_counter_collection.csv $out/${tag}_WRITE_SIZE/p_counter_collection.csv $out/${tag}_traffic.json
rm -f $out/${tag}_FETCH_SIZE/p_counter_collection.csv $out/${tag}_WRITE_SIZE/p_counter_collection.csv
