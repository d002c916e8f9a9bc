#!/bin/bash
out=$PWD/gpurun_out
python -m pytest tests/test_gpu_edge_dp.py -q -m gpu -rf > $out/r3l_tests.log 2>&1; tail -8 $out/r3l_tests.log | cut -c1-400
