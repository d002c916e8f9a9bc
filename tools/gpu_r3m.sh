#!/bin/bash
repo=$PWD; out=$repo/gpurun_out; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r3m_prof_train -o train -- python $repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-extras > $out/r3m_prof_train.log 2>&1
cd $repo
find $out/r3m_prof_train -name "*.db" -delete
python tools/timeline.py $out/r3m_prof_train/*/train_kernel_trace.csv > $out/r3m_timeline.txt 2>&1 || true
find $out/r3m_prof_train -name "*kernel_trace.csv" -delete
ls -R $out/r3m_prof_train | head
