#!/bin/bash
# kernel trace of the training step (current tree): per-kernel timeline of one step + A/B against the round-2 library
repo=$PWD; out=$repo/gpurun_out; export TMPDIR=/tmp
for arm in "B2S_LIB_PATH=$repo/tools/bin/libb2s_r02.so" "B2S_X=0" "B2S_LIB_PATH=$repo/tools/bin/libb2s_r02.so" "B2S_X=0"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms"
done
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/r3e_trace -o t -- python $repo/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/r3e_trace.log 2>&1
cd $repo
f=$(find $out/r3e_trace -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f --list > $out/r3e_timeline.txt 2>&1
tail -70 $out/r3e_timeline.txt
find $out/r3e_trace -name "*.csv" -delete; find $out/r3e_trace -name "*.db" -delete
