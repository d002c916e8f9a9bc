#!/bin/bash
# persistent multi-round GEMM kernel: correctness, then A/B of the step (lab build: B2S_LAB_GEMM_PERSIST = 0 plain kernel, 1 tickets, 2 static tile walk)
repo=$PWD; out=$repo/gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "persistent or gemm_forms or gemm_256" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 > $out/r5p_tests.log
cat $out/r5p_tests.log
export B2S_LIB_PATH=$repo/tools/bin/libb2s_hip_lab.so
for v in 0 2 1; do echo persist=$v; B2S_LAB_GEMM_PERSIST=$v python tools/gemm_persist_lab.py 2>&1 | grep "us " | grep -v "relu "; done
for r in 1 2 3; do
  for v in ${ARMS:-0 2 1}; do
    echo -n "persist=$v " >> $out/r5p_ab.txt
    B2S_LAB_GEMM_PERSIST=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline-pass --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'])" >> $out/r5p_ab.txt
  done
done
cat $out/r5p_ab.txt
# the backward pass with CUs held (what a collective's channels do), 256 x 128 tiles as the data-parallel trainer selects them
for v in ${ARMS:-0 2 1}; do echo "persist=$v, CUs held during the backward pass"; B2S_LAB_GEMM_PERSIST=$v B2S_GEMM256_NB=4 timeout 300 python tools/cu_loss.py bwd 2>&1 | grep "held CUs"; done | tee $out/r5p_cu_loss.txt
