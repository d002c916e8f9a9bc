#!/bin/bash
out=$PWD/gpurun_out
python -m pytest tests/test_gpu_trainer_state.py tests/test_gpu_ops.py -q -m gpu -rf -k "abandoned or tile_policy or split_optimizer or stage_hook" > $out/r3p_tests.log 2>&1; tail -8 $out/r3p_tests.log | cut -c1-300
