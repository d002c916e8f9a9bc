#!/bin/bash
# round 3: attention variants lab, parity suite on the current tree, bench (train leg only) A/B over the attention variants
out=$PWD/gpurun_out; mkdir -p $out
tools/bin/attn_lab2 > $out/r3b_attn_lab.txt 2>&1; cat $out/r3b_attn_lab.txt
python -m pytest tests -q -m gpu > $out/r3b_tests.log 2>&1; tail -8 $out/r3b_tests.log
for arm in "B2S_ATTN_RB=1 B2S_ATTN_KBW=1" "B2S_ATTN_RB=2 B2S_ATTN_KBW=1" "B2S_ATTN_RB=2 B2S_ATTN_KBW=2" "B2S_ATTN_RB=1 B2S_ATTN_KBW=1" "B2S_ATTN_RB=2 B2S_ATTN_KBW=1"; do
  ms=$(env $arm python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a $out/r3b_ab.txt
done
