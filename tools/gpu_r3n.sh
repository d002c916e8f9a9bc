#!/bin/bash
out=$PWD/gpurun_out; : > $out/r3n.txt
for r in 1 2; do
B2S_ADAM_V1=1 python tools/adam_lab.py 2>&1 | tail -1 | tee -a $out/r3n.txt
python tools/adam_lab.py 2>&1 | tail -1 | tee -a $out/r3n.txt
done
python -m pytest tests -q -m gpu -x -k "trainer or adam or checkpoint or lj_shape" 2>&1 | tail -3 | tee -a $out/r3n.txt
