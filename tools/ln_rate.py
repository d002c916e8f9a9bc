"""LayerNorm kernels with the GPU to themselves (C-ABI ops, M = 8148, D = 768, bf16): run under rocprofv3 --kernel-trace --stats."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))
from b2s_hip import lib as L
lib = L.load()
M, D = 8148, 768
x = torch.randn(M, D, device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
y = torch.empty(M, D, device="cuda", dtype=torch.bfloat16); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
dy = torch.randn(M, D, device="cuda").to(torch.bfloat16); dx = torch.empty(M, D, device="cuda")
dg = torch.empty(D, device="cuda"); db = torch.empty(D, device="cuda")
for _ in range(30):
    L.check(lib.b2s_layernorm_forward(1, L.ptr(x), L.ptr(g), L.ptr(b), L.ptr(y), L.ptr(mean), L.ptr(rstd), M, D, 1e-6, L.stream()))
    L.check(lib.b2s_layernorm_backward(1, L.ptr(dy), L.ptr(x), L.ptr(g), L.ptr(mean), L.ptr(rstd), L.ptr(dx), L.ptr(dg), L.ptr(db), M, D, L.stream()))
torch.cuda.synchronize()
print("forward: %.1f MB per launch, backward (no residual-gradient input, no second output): %.1f MB per launch" % (M * D * 6 / 1e6, M * D * 10 / 1e6))
