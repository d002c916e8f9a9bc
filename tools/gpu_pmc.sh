#!/bin/bash
# HBM traffic of the training step from the L2's memory-side counters + MFMA utilisation: three separate rocprofv3 --pmc passes (never
# combined with trace domains other than the kernel trace).  Usage (through gpurun): bash tools/gpu_pmc.sh <tag>
tag=${1:-pmc}
repo=$PWD
out=$repo/gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_$c -o p -- python $repo/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline-pass --no-extras > $out/${tag}_$c.log 2>&1
  rm -f $out/${tag}_$c/*kernel_trace.csv
done
# matrix-pipe utilisation: MFMA-busy cycles against the kernel's active cycles (third pass, SQ + GRBM counters only)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_MFMA -o p -- python $repo/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline-pass --no-extras > $out/${tag}_MFMA.log 2>&1
rm -f $out/${tag}_MFMA/*kernel_trace.csv
cd $repo
python tools/pmc_traffic.py $out/${tag}_FETCH_SIZE/p_counter_collection.csv $out/${tag}_WRITE_SIZE/p_counter_collection.csv $out/${tag}_traffic.json $out/${tag}_MFMA/p_counter_collection.csv
rm -f $out/${tag}_MFMA/p_counter_collection.csv $out/${tag}_FETCH_SIZE/p_counter_collection.csv $out/${tag}_WRITE_SIZE/p_counter_collection.csv
