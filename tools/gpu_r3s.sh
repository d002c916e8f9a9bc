#!/bin/bash
out=$PWD/gpurun_out
python -m pytest tests -q -m gpu -rf > $out/r3s_tests.log 2>&1; tail -6 $out/r3s_tests.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
