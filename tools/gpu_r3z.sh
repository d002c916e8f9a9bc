#!/bin/bash
out=$PWD/gpurun_out
python -m pytest tests -q -m gpu -x -k "pack or rccl or workload or data_parallel or wrappers or ops or trainer" > $out/r3z_tests.log 2>&1; tail -2 $out/r3z_tests.log | cut -c1-200
python - <<'PY'
import sys, torch
sys.path.insert(0, "few-shot-transformer-tts_amd")
from b2s_hip import lib as L
lib = L.load()
n = 83477440
src = torch.randn(n, device="cuda"); dst = torch.empty(n, device="cuda", dtype=torch.bfloat16); back = torch.empty(n, device="cuda")
for name, f in (("pack", lambda: lib.b2s_pack_bf16(src.data_ptr(), dst.data_ptr(), n, L.stream())), ("unpack", lambda: lib.b2s_unpack_bf16(dst.data_ptr(), back.data_ptr(), n, L.stream()))):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("%s of %d elements: %.1f us = %.2f TB/s" % (name, n, us, n * 6 / us / 1e6))
assert torch.equal(dst.float(), src.to(torch.bfloat16).float()) and torch.equal(back, dst.float())
print("pack == torch bf16 rounding, unpack exact")
PY
