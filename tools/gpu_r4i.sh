#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dp_race.py tests/test_gpu_edge_dp.py tests/test_gpu_wrappers.py tests/test_gpu_trainer_state.py -q 2>&1 | tail -25 > gpurun_out/r4i_dp.log
python -m pytest "tests/test_gpu_fullsize.py::test_c3_per_rank_workload_through_rccl_exchange_path" -q 2>&1 | tail -8 >> gpurun_out/r4i_dp.log
export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1
for r in 1 2; do
  for arm in plain dp_wire dp_unpack; do
    case $arm in plain) e="";; dp_wire) e="B2S_FORCE_DP=1 B2S_GRAD_PAYLOAD=bf16 MASTER_PORT=2951$r";; dp_unpack) e="B2S_FORCE_DP=1 B2S_GRAD_PAYLOAD=bf16 B2S_ADAM_FROM_WIRE=0 MASTER_PORT=2952$r";; esac
    ms=$(env $e python bench.py --no-cpu-baseline --no-roofline-pass --no-extras --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
    echo "round $r $arm $ms" | tee -a gpurun_out/r4i_dp_ab.txt
  done
done
