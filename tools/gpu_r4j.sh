#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/r4j.txt
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 B2S_FORCE_DP=1
p=29600
for arm in "-" "B2S_GRAD_PAYLOAD=fp32" "B2S_ENC_OVERLAP=0" "B2S_ENC_FUSED=0" "B2S_DW_TAIL_LAYERS=0" "B2S_ADAM_FROM_WIRE=0" "B2S_DX_BF16=0" "GPU_MAX_HW_QUEUES=16" "B2S_NO_HOOK_STREAM=1"; do
  p=$((p+1)); envs="MASTER_PORT=$p"; [ "$arm" != "-" ] && envs="$envs $arm"
  ms=$(env $envs python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 30 --warmup 6 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
  echo "[$arm] $ms" | tee -a gpurun_out/r4j.txt
done
