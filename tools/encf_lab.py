"""Isolated timing of the fused encoder kernels (op-level C ABI), B=14 S=114 by default: us per launch, back to back on an idle GPU."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "few-shot-transformer-tts_amd"))
from b2s_hip import lib as L
lib = L.load()
B, S = int(os.environ.get("LAB_B", 14)), int(os.environ.get("LAB_S", 114))
p = float(os.environ.get("LAB_P", 0.1))
M, D, F, H = B * S, 512, 2048, 8
dev = "cuda"
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
h, Wqkv, Wo, W1, W2 = bf(M, D) * 20, bf(3 * D, D), bf(D, D), bf(F, D), bf(D, F)
WoT, WqkvT, W1T, W2T = Wo.t().contiguous(), Wqkv.t().contiguous(), W1.t().contiguous(), W2.t().contiguous()
klen = torch.full((B,), S, dtype=torch.int32, device=dev)
qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev); ctx = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B * H, S, device=dev); f = torch.empty(M, F, dtype=torch.bfloat16, device=dev); dz = torch.empty_like(f); dqkv = torch.empty_like(qkv)
x = torch.randn(M, D, device=dev); xo = torch.empty_like(x); hh = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev); g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
dx = torch.zeros(M, D, device=dev); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev); ws = torch.empty(768 * 1024, device=dev)
dy2 = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
def run(name, fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-28s %7.2f us" % (name, e0.elapsed_time(e1) / n * 1e3))
for sb in (0, 1):
    sl = torch.zeros(16 * M * D, dtype=torch.bfloat16 if sb else torch.float32, device=dev)
    print("slab_bf16 =", sb, " B =", B, " S =", S, " p =", p)
    run("attn fwd", lambda: L.check(lib.b2s_encf_attention_forward(h.data_ptr(), Wqkv.data_ptr(), Wo.data_ptr(), klen.data_ptr(), B, S, p, 1, 2, qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), sl.data_ptr(), sb, None)))
    run("attn bwd", lambda: L.check(lib.b2s_encf_attention_backward(h.data_ptr(), qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), WoT.data_ptr(), WqkvT.data_ptr(), klen.data_ptr(), B, S, p, 1, 2, dqkv.data_ptr(), sl.data_ptr(), sb, None)))
    run("ffn fwd", lambda: L.check(lib.b2s_encf_ffn_sublayer(0, h.data_ptr(), W1.data_ptr(), W2.data_ptr(), f.data_ptr(), None, B, S, p, 1, 3, sl.data_ptr(), sb, None)))
    run("ffn bwd", lambda: L.check(lib.b2s_encf_ffn_sublayer(1, h.data_ptr(), W2T.data_ptr(), W1T.data_ptr(), f.data_ptr(), dz.data_ptr(), B, S, p, 1, 3, sl.data_ptr(), sb, None)))
    for ns in (8,):
        run("reduce+LN fwd ns=%d" % ns, lambda: L.check(lib.b2s_encf_reduce_layernorm_forward(x.data_ptr(), sl.data_ptr(), ns, sb, p, 1, 4, g.data_ptr(), b.data_ptr(), xo.data_ptr(), hh.data_ptr(), None, 0, mean.data_ptr(), rstd.data_ptr(), M, None)))
        run("reduce+LN bwd ns=%d" % ns, lambda: L.check(lib.b2s_encf_reduce_layernorm_backward(sl.data_ptr(), ns, sb, xo.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), dy2.data_ptr(), p, 1, 5, M, None)))
run("transpose 1536x512", lambda: L.check(lib.b2s_transpose_bf16(Wqkv.data_ptr(), WqkvT.data_ptr(), 3 * D, D, None)))
