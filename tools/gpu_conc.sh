#!/bin/bash
# Does the step get cheaper per utterance when two half-batch chains share the GPU?  (1) B=7 / 14 / 28 alone; (2) two B=7
# processes at the same time.
out=gpurun_out/conc_r2.txt; : > $out
run() { python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])"; }
for b in 7 14 28; do echo "alone B=$b: $(run --batch $b)" | tee -a $out; done
(run --batch 7 > /tmp/c1.txt) & (run --batch 7 > /tmp/c2.txt) & wait
echo "two concurrent B=7 processes: $(cat /tmp/c1.txt) | $(cat /tmp/c2.txt)" | tee -a $out
(run --batch 14 > /tmp/c1.txt) & (run --batch 14 > /tmp/c2.txt) & wait
echo "two concurrent B=14 processes: $(cat /tmp/c1.txt) | $(cat /tmp/c2.txt)" | tee -a $out
