#!/bin/bash
out=$PWD/gpurun_out
python -m pytest tests -q -m gpu -rf > $out/r3v_tests.log 2>&1; tail -3 $out/r3v_tests.log | cut -c1-300
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('train ms_per_step', d['ms_per_step'])"
done | tee $out/r3v_bench.txt
B2S_LIB_PATH=$PWD/tools/bin/libb2s_r02.so B2S_ENC_OVERLAP=0 python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>&1 | tail -2 | cut -c1-200 | tee -a $out/r3v_bench.txt
