"""Op-level parity of the HIP kernels (through the C ABI) against plain fp32/fp64 torch-CPU math.

fp32 mode uses the exact-fp32 MFMA: tolerance 2e-5 relative.  bf16 mode: operands are rounded to bf16 in the
reference too, so only accumulation order / output rounding differ: tolerance 1e-2 relative (bf16 has 8 bits).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, bf16_round, to_dev_compute, from_dev_compute, relerr, report  # noqa: E402

TOL = {0: 2e-5, 1: 1e-2}


def _ops():
    from b2s_hip import ops, lib
    lib.load()
    return ops, lib


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 80), (77, 40, 264), (16, 8, 8), (513, 257, 1032)])
def test_gemm_forms(dtype, ta, tb, M, N, K):
    ops, lib = _ops()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + dtype)
    # asymmetric operands (transpose-detecting)
    A = torch.randn(M, K, generator=g) + torch.arange(K)[None, :] * 0.01
    B = torch.randn(N, K, generator=g) - torch.arange(N)[:, None] * 0.01
    if dtype:
        A, B = bf16_round(A), bf16_round(B)
    ref = A.double() @ B.double().t()
    # pad leading dims to multiples of 8 where the contiguous dim is M or N
    def stored(X, trans):   # returns (device tensor, ld)
        if not trans:
            return to_dev_compute(X, dtype), X.shape[1]
        Xt = X.t().contiguous()                       # [K, rows]
        ld = (Xt.shape[1] + 7) // 8 * 8
        buf = torch.zeros(Xt.shape[0], ld)
        buf[:, :Xt.shape[1]] = Xt
        return to_dev_compute(buf, dtype), ld
    Ad, lda = stored(A, ta)
    Bd, ldb = stored(B, tb)
    C = ops.gemm(dtype, Ad, Bd, M, N, K, trans_a=ta, trans_b=tb, lda=lda, ldb=ldb)
    torch.cuda.synchronize()
    err = relerr(C.cpu(), ref)
    assert err < TOL[dtype], report("gemm", C.cpu(), ref)


@pytest.mark.parametrize("dtype", [0, 1])
def test_gemm_batched_heads_and_epilogue(dtype):
    ops, lib = _ops()
    g = torch.Generator().manual_seed(5)
    B_, H, L, dh = 3, 2, 37, 32
    D = H * dh
    q = torch.randn(B_, L, D, generator=g); k = torch.randn(B_, L, D, generator=g)
    if dtype:
        q, k = bf16_round(q), bf16_round(k)
    ref = torch.einsum("blhd,bmhd->bhlm", q.view(B_, L, H, dh).double(), k.view(B_, L, H, dh).double())
    ldp = (L + 7) // 8 * 8
    out = torch.zeros(B_ * H, L, ldp, device=DEV)
    ops.gemm(dtype, to_dev_compute(q, dtype), to_dev_compute(k, dtype), L, L, dh, lda=D, ldb=D, ldc=ldp, out=out,
             batch=B_ * H, batch_inner=H, a_bs=(L * D, dh), b_bs=(L * D, dh), c_bs=(H * L * ldp, L * ldp))
    torch.cuda.synchronize()
    got = out.cpu().view(B_, H, L, ldp)[..., :L]
    assert relerr(got, ref) < TOL[dtype], report("batched", got, ref)
    # epilogue: bias + relu + residual, compute-dtype output
    M, N, K = 70, 48, 64
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) * 0.2
    bias = torch.randn(N, generator=g); res = torch.randn(M, N, generator=g)
    if dtype:
        A, W = bf16_round(A), bf16_round(W)
    ref = torch.relu(A.double() @ W.double().t() + bias.double()) + res.double()
    C = ops.gemm(dtype, to_dev_compute(A, dtype), to_dev_compute(W, dtype), M, N, K, bias=bias.to(DEV), relu=True,
                 residual=res.to(DEV))
    torch.cuda.synchronize()
    assert relerr(C.cpu(), ref) < TOL[dtype], report("epilogue", C.cpu(), ref)


@pytest.mark.parametrize("dtype,M,N,K", [(1, 64, 768, 3072), (1, 37, 520, 2048), (0, 64, 768, 3072), (1, 64, 768, 768)])
def test_decode_step_gemm_in_place_residual(dtype, M, N, K):
    """The decode step's ffn-out form: <= 64 rows, h += x W^T + b in place (fp32 h).  bf16 with K >= 2048 takes the
    weight-streaming kernel's K-split-over-workgroups path (atomic adds into h); the other rows pin the single-pass path."""
    ops, lib = _ops()
    g = torch.Generator().manual_seed(M + N + K + dtype)
    X = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) * 0.05
    h = torch.randn(M, N, generator=g); bias = torch.randn(N, generator=g)
    if dtype:
        X, W = bf16_round(X), bf16_round(W)
    ref = h.double() + X.double() @ W.double().t() + bias.double()[None, :]
    hd = h.to(DEV).contiguous()
    out = ops.gemm(dtype, to_dev_compute(X, dtype), to_dev_compute(W, dtype), M, N, K, out=hd, c_fp32=True, bias=bias.to(DEV), residual=hd)
    torch.cuda.synchronize()
    assert out.data_ptr() == hd.data_ptr()
    assert relerr(hd.cpu(), ref) < TOL[dtype], report("in-place residual", hd.cpu(), ref)


@pytest.mark.parametrize("with_len", [True, False])
def test_conv_gather_gemm_aligned_channels(with_len):
    """The postnet's 512-channel layers: channel count a multiple of the GEMM's 64-deep K step and > 128 rows -> the
    aligned gather of the 256-row tile kernel (producer waves, one compare per DMA instruction).  Ragged lengths with a
    tile boundary inside an utterance; with_len=False is the backward-data form (no length mask on the gathered operand)."""
    ops, lib = _ops()
    g = torch.Generator().manual_seed(19)
    B_, T, Cin, Cout = 3, 181, 128, 80
    lens = torch.tensor([181, 97, 3], dtype=torch.int32)
    x = bf16_round(torch.randn(B_, T, Cin, generator=g)); w = bf16_round(torch.randn(Cout, Cin, 5, generator=g) * 0.1)
    mask = (torch.arange(T)[None, :] < lens[:, None]).float() if with_len else torch.ones(B_, T)
    ref = torch.nn.functional.conv1d((x * mask[..., None]).transpose(1, 2).double(), w.double(), None, 1, 2).transpose(1, 2)
    wf = w.permute(0, 2, 1).reshape(Cout, 5 * Cin).contiguous()
    y = ops.gemm(1, to_dev_compute(x.reshape(B_ * T, Cin), 1), to_dev_compute(wf, 1), B_ * T, Cout, 5 * Cin,
                 lda=Cin, ldb=5 * Cin, conv_cin_a=Cin, conv_T=T, conv_len=lens.to(DEV) if with_len else None)
    torch.cuda.synchronize()
    got = y.cpu().view(B_, T, Cout)
    assert relerr(got, ref) < TOL[1], report("conv aligned", got, ref)


@pytest.mark.parametrize("dtype", [0, 1])
def test_conv_gather_gemm(dtype):
    """impute + Conv1d(k=5, pad=2) as an implicit GEMM (tacotron.py:84-85)."""
    ops, lib = _ops()
    g = torch.Generator().manual_seed(9)
    B_, T, Cin, Cout = 3, 29, 16, 24
    lens = torch.tensor([29, 17, 5], dtype=torch.int32)
    x = torch.randn(B_, T, Cin, generator=g); w = torch.randn(Cout, Cin, 5, generator=g) * 0.2
    if dtype:
        x, w = bf16_round(x), bf16_round(w)
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()
    ref = torch.nn.functional.conv1d((x * mask[..., None]).transpose(1, 2).double(), w.double(), None, 1, 2).transpose(1, 2)
    wf = w.permute(0, 2, 1).reshape(Cout, 5 * Cin).contiguous()            # [co][j*Cin+ci]
    y = ops.gemm(dtype, to_dev_compute(x.reshape(B_ * T, Cin), dtype), to_dev_compute(wf, dtype), B_ * T, Cout, 5 * Cin,
                 lda=Cin, ldb=5 * Cin, conv_cin_a=Cin, conv_T=T, conv_len=lens.to(DEV))
    torch.cuda.synchronize()
    got = y.cpu().view(B_, T, Cout)
    assert relerr(got, ref) < TOL[dtype], report("conv", got, ref)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("D", [64, 512, 768])
def test_layernorm_fwd_bwd(dtype, D):
    ops, lib = _ops()
    g = torch.Generator().manual_seed(D)
    M = 45
    x = (torch.randn(M, D, generator=g) * 2 + 0.5).requires_grad_(True)
    w = (1 + 0.1 * torch.randn(D, generator=g)).requires_grad_(True)
    b = (0.1 * torch.randn(D, generator=g)).requires_grad_(True)
    go = torch.randn(M, D, generator=g)
    if dtype:
        go = bf16_round(go)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-6)
    ref.backward(go.double())
    xd = x.detach().to(DEV).requires_grad_(True); wd = w.detach().to(DEV).requires_grad_(True); bd = b.detach().to(DEV).requires_grad_(True)
    y = ops.layernorm(xd, wd, bd, 1e-6, dtype)
    y.backward(go.to(DEV))
    torch.cuda.synchronize()
    tol = 1e-5 if not dtype else 1e-2
    assert relerr(y.detach().cpu(), ref.detach()) < tol, report("ln y", y.detach().cpu(), ref.detach())
    assert relerr(xd.grad.cpu(), x.grad) < 2e-5 * (1 if not dtype else 500), report("ln dx", xd.grad.cpu(), x.grad)
    assert relerr(wd.grad.cpu(), w.grad) < 2e-5 * (1 if not dtype else 500), report("ln dg", wd.grad.cpu(), w.grad)
    assert relerr(bd.grad.cpu(), b.grad) < 2e-5 * (1 if not dtype else 500), report("ln db", bd.grad.cpu(), b.grad)


def _attn_ref(q, k, v, H, mask_mode, klen):
    B_, Lq, C = q.shape
    Lk = k.shape[1]
    dh = C // H
    qh = q.view(B_, Lq, H, dh).permute(0, 2, 1, 3); kh = k.view(B_, Lk, H, dh).permute(0, 2, 1, 3)
    vh = v.view(B_, Lk, H, dh).permute(0, 2, 1, 3)
    logits = (qh * dh ** -0.5) @ kh.transpose(2, 3)
    if mask_mode & 1:
        m = torch.arange(Lk)[None, :] < klen[:, None]
        logits = logits + ((1.0 - m.double()) * -1e20)[:, None, None, :]
    if mask_mode & 2:
        logits = logits + (torch.triu(torch.ones(Lq, Lk, dtype=torch.float64), 1) * -1e20)[None, None]
    p = torch.softmax(logits, -1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B_, Lq, C), p


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("dh,mask_mode,Lq,Lk", [(32, 1, 21, 13), (64, 2, 40, 40), (96, 1, 150, 33), (96, 2, 130, 130), (64, 1, 70, 200),
                                                   # resident-key kernels (no causal mask, Lk <= 128, more than one query tile): the benchmark's
                                                   # encoder-decoder shape, a full 128-key image, 65 keys (one valid key in the second tile), no mask at all
                                                   (96, 1, 582, 114), (64, 1, 200, 128), (32, 1, 130, 65), (96, 0, 321, 100)])
def test_attention_core_fwd_bwd(fused, dtype, dh, mask_mode, Lq, Lk):
    ops, lib = _ops()
    g = torch.Generator().manual_seed(dh + Lq)
    B_, H = 2, 2
    C = H * dh
    q = torch.randn(B_, Lq, C, generator=g); k = torch.randn(B_, Lk, C, generator=g); v = torch.randn(B_, Lk, C, generator=g)
    go = torch.randn(B_, Lq, C, generator=g)
    if dtype:
        q, k, v, go = bf16_round(q), bf16_round(k), bf16_round(v), bf16_round(go)
    klen = torch.tensor([Lk, max(1, Lk - 5)], dtype=torch.int32)
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, pref = _attn_ref(qr, kr, vr, H, mask_mode, klen)
    ref.backward(go.double())
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    out, probs = ops.attention_core(qd, kd, vd, H, mask_mode, klen.to(DEV) if mask_mode & 1 else None, None, 0.0, 0, dtype, fused=fused)
    out.backward(go.to(DEV))
    torch.cuda.synchronize()
    tol = 3e-5 if not dtype else 2e-2
    assert relerr(out.detach().cpu(), ref.detach()) < tol, report("attn out", out.detach().cpu(), ref.detach())
    assert relerr(probs.cpu(), pref.detach()) < tol, report("attn probs", probs.cpu(), pref.detach())
    for name, a, b in (("dq", qd.grad, qr.grad), ("dk", kd.grad, kr.grad), ("dv", vd.grad, vr.grad)):
        assert relerr(a.cpu(), b) < (1e-4 if not dtype else 3e-2), report("attn " + name, a.cpu(), b)


def test_attention_dense_bias_matches_mask():
    """A reference-style dense -1e20 bias tensor gives the same result as the in-kernel length mask."""
    ops, lib = _ops()
    g = torch.Generator().manual_seed(3)
    B_, H, dh, Lq, Lk = 2, 2, 32, 9, 12
    C = H * dh
    q = torch.randn(B_, Lq, C, generator=g).to(DEV); k = torch.randn(B_, Lk, C, generator=g).to(DEV); v = torch.randn(B_, Lk, C, generator=g).to(DEV)
    klen = torch.tensor([12, 7], dtype=torch.int32)
    dense = ((1.0 - (torch.arange(Lk)[None, :] < klen[:, None]).float()) * -1e20)[:, None, None, :]
    a, pa = ops.attention_core(q, k, v, H, 1, klen.to(DEV), None, 0.0, 0, 0, fused=False)
    b, pb = ops.attention_core(q, k, v, H, 0, None, dense.to(DEV), 0.0, 0, 0)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(pa, pb)


@pytest.mark.parametrize("Lk", [300, 256, 200, 140, 131, 114, 70])          # (<= 128: the resident-key kernels)
@pytest.mark.parametrize("dtype", [0, 1])
def test_fused_attention_matches_materialised_with_dropout(dtype, Lk):
    """Same seed -> the fused kernels and the GEMM+softmax path draw the same dropout mask: outputs and gradients agree.
    (Even and odd key counts: the row stride of the mask index.)"""
    ops, lib = _ops()
    g = torch.Generator().manual_seed(17)
    B_, H, dh, Lq = 2, 2, 96, 75
    C = H * dh
    q = torch.randn(B_, Lq, C, generator=g); k = torch.randn(B_, Lk, C, generator=g); v = torch.randn(B_, Lk, C, generator=g)
    go = torch.randn(B_, Lq, C, generator=g)
    klen = torch.tensor([Lk, Lk - 17], dtype=torch.int32).to(DEV)
    res = []
    for fused in (True, False):
        qd, kd, vd = (t.clone().to(DEV).requires_grad_(True) for t in (q, k, v))
        out, _ = ops.attention_core(qd, kd, vd, H, 1, klen, None, 0.3, 4242, dtype, fused=fused)
        out.backward(go.to(DEV))
        torch.cuda.synchronize()
        res.append((out.detach().cpu(), qd.grad.cpu(), kd.grad.cpu(), vd.grad.cpu()))
    tol = 1e-4 if not dtype else 3e-2
    for a, b, name in zip(res[0], res[1], ("out", "dq", "dk", "dv")):
        assert relerr(a, b) < tol, report("fused vs materialised " + name, a, b)


@pytest.mark.parametrize("N", [96, 100, 72])
def test_gemm_epilogue_dropout_equals_exported_mask(N):
    """The GEMM epilogue's dropout mask (eight elements per lane and pass) is the one the exported mask op reports element by element.
    C = A B^T with every product equal to 1 shows the mask directly."""
    ops, lib = _ops()
    l = lib.load()
    M, K, p, seed = 300, 64, 0.25, 99
    for dtype in (0, 1):
        A = ops.to_compute(torch.full((M, K), 1.0 / K, device=DEV), dtype)
        Bm = ops.to_compute(torch.ones(N, K, device=DEV), dtype)
        C = ops.gemm(dtype, A, Bm, M, N, K, trans_b=False, drop_p=p, seed=seed)
        m = torch.empty(M * N, dtype=torch.uint8, device=DEV)
        lib.check(l.b2s_dropout_mask(p, seed, 7, lib.ptr(m), M * N, lib.stream()))
        torch.cuda.synchronize()
        assert torch.equal((C.reshape(-1) > 0.5).to(torch.uint8), m), (dtype, N)
        kept = C.reshape(-1)[m.bool()]
        assert torch.allclose(kept, torch.full_like(kept, 1.0 / (1 - p)), rtol=1e-2)


def test_dropout_rng_statistics():
    """Keep-rate of the counter-based RNG matches 1-p; different ops/seeds give different masks; same key replays."""
    ops, lib = _ops()
    l = lib.load()
    n = 1 << 20
    for p in (0.1, 0.5):
        m = torch.empty(n, dtype=torch.uint8, device=DEV)
        lib.check(l.b2s_dropout_mask(p, 1234, 5, lib.ptr(m), n, lib.stream()))
        m2 = torch.empty_like(m); m3 = torch.empty_like(m)
        lib.check(l.b2s_dropout_mask(p, 1234, 5, lib.ptr(m2), n, lib.stream()))
        lib.check(l.b2s_dropout_mask(p, 1234, 6, lib.ptr(m3), n, lib.stream()))
        torch.cuda.synchronize()
        keep = m.float().mean().item()
        assert abs(keep - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-4, keep
        assert torch.equal(m, m2)
        agree = (m == m3).float().mean().item()
        assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 5e-3, agree
        # no obvious serial correlation
        c = (m[1:] & m[:-1]).float().mean().item()
        assert abs(c - (1 - p) ** 2) < 5e-3, c


def test_dropout_mask_matches_host_restatement():
    """Device masks == oracle/rng.py (the restatement whose statistics tests/test_host_logic.py checks), bit for bit, incl. odd starts."""
    from oracle import rng
    ops, lib = _ops()
    l = lib.load()
    for p, seed, op, n in ((0.1, 1234, 5, 200003), (0.5, (7 << 40) + 3, 4099, 65536), (0.9999, 1, 2, 4097), (1e-6, 1, 2, 4097)):
        m = torch.empty(n, dtype=torch.uint8, device=DEV)
        lib.check(l.b2s_dropout_mask(p, seed, op, lib.ptr(m), n, lib.stream()))
        torch.cuda.synchronize()
        assert np.array_equal(m.cpu().numpy().astype(bool), rng.keep_mask(p, seed, op, n)), (p, seed, op)


def test_attention_dropout_mask_matches_host_restatement():
    """The training kernels' attention-weight masks (row seeds + key quads: csrc/b2s_common.h: b2s_keep_w) == oracle/rng.py: keep_mask_attn,
    bit for bit, for key counts that are and are not multiples of four."""
    from oracle import rng
    ops, lib = _ops()
    l = lib.load()
    for p, seed, op, rows, Lk in ((0.1, 1234, 8324, 997, 582), (0.1, (7 << 40) + 3, 4162, 513, 113), (0.5, 5, 6, 64, 1), (0.9999, 1, 2, 33, 7)):
        m = torch.empty(rows * Lk, dtype=torch.uint8, device=DEV)
        lib.check(l.b2s_dropout_mask_attn(p, seed, op, lib.ptr(m), rows, Lk, lib.stream()))
        torch.cuda.synchronize()
        assert np.array_equal(m.cpu().numpy().astype(bool).reshape(rows, Lk), rng.keep_mask_attn(p, seed, op, rows, Lk)), (p, seed, op)


@pytest.mark.parametrize("nb", ["3", "4"])
def test_gemm_256_tile_kernel_all_forms(nb):
    """The 256x128 / 256x96 tile kernel (gemm_glds256.hip) is normally chosen for M > 128 with a shape-dependent tile width;
    force it for every non-batched bf16 GEMM (B2S_GEMM256_MIN_M=1) and re-run the GEMM / conv / bf16 model tests through it, both tile widths."""
    import os
    import subprocess
    import sys
    if os.environ.get("B2S_GEMM256_MIN_M"):
        pytest.skip("inner run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, B2S_GEMM256_MIN_M="1", B2S_GEMM256_NB=nb)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_ops.py", "tests/test_gpu_model.py",
                        "-k", "test_gemm_forms or test_gemm_batched or test_conv_gather or bf16 or test_decode"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("policy", [0, 4])
@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("M,N,K", [(3000, 3072, 384), (2900, 2304, 64), (3333, 4608, 832), (1000, 768, 512), (2040, 1536, 448)])
def test_gemm_persistent_multi_round_kernel(policy, tb, M, N, K):
    """Launches with a compute-dtype output and K >= 384 -- several rounds of tiles (> 256 tiles of 256 x 96 / 128) or a single one -- take the persistent kernel (gemm_glds256.hip: one workgroup walks
    several tiles, the operand ring rolls across the tile boundary, the epilogue is staged outside the ring).  Its compute-dtype results -- plain,
    ReLU, ReLU + dropout -- must be BIT-identical to the plain kernel's arithmetic: the fp32 products of the same GEMM computed in row slices of one
    round each (plain kernel, fp32 output), then ReLU / exported dropout mask / one rounding to bf16 on the host side."""
    ops, lib = _ops()
    l = lib.load()
    torch.manual_seed(M + N + K + int(tb))
    A = ops.to_compute(torch.randn(M, K, device=DEV), 1)
    Bm = ops.to_compute(torch.randn(K, N, device=DEV) if tb else torch.randn(N, K, device=DEV), 1)
    lib.check(l.b2s_gemm_set_tile_policy(policy))
    try:
        acc = torch.empty(M, N, device=DEV)
        step = 512                                     # 2 row panels x <= 48 column panels: one round, plain kernel
        for r0 in range(0, M, step):
            r1 = min(M, r0 + step)
            acc[r0:r1] = ops.gemm(1, A[r0:r1].contiguous(), Bm, r1 - r0, N, K, trans_b=tb, c_fp32=True)
        p, seed = 0.2, 4242
        mask = torch.empty(M * N, dtype=torch.uint8, device=DEV)
        lib.check(l.b2s_dropout_mask(p, seed, 7, lib.ptr(mask), M * N, lib.stream()))
        keep = mask.reshape(M, N).bool()
        cases = {"plain": (dict(), acc),
                 "relu": (dict(relu=True), acc.clamp_min(0)),
                 "relu+dropout": (dict(relu=True, drop_p=p, seed=seed),
                                  torch.where(keep, acc.clamp_min(0) * float(np.float32(1) / (np.float32(1) - np.float32(p))), torch.zeros_like(acc)))}
        # (a three-round launch first: it leaves real list positions in the workgroups' LDS ticket queues -- a one-round launch that read its
        # queue instead of stopping after its only tile would walk those)
        big = ops.to_compute(torch.zeros(2048, 384, device=DEV), 1)
        ops.gemm(1, big, ops.to_compute(torch.zeros(9216, 384, device=DEV), 1), 2048, 9216, 384, trans_b=False, c_fp32=False)
        for what, (kw, want) in cases.items():
            got = ops.gemm(1, A, Bm, M, N, K, trans_b=tb, c_fp32=False, **kw)
            torch.cuda.synchronize()
            want16 = want.to(torch.bfloat16).view(torch.int16)
            bad = int((got != want16).sum())
            assert bad == 0, (what, policy, tb, M, N, K, bad)
    finally:
        lib.check(l.b2s_gemm_set_tile_policy(0))


@pytest.mark.parametrize("slabs", [True, False])
def test_splitk_weight_gradient_gemms_on_two_streams(slabs):
    """Split-K slab workspaces belong to the caller, one per stream (GemmArgs::ws; the library owns none): two different
    weight-gradient GEMMs (TN form, K = 8192 split 8 ways) issued back to back on two streams, many rounds, both results
    right every round.  slabs=False: the fp32-atomic accumulation path."""
    import ctypes as C
    ops, lib = _ops()
    L = lib
    l = L.load()
    g = torch.Generator().manual_seed(11)
    K = 8192
    shapes = [(768, 640), (512, 768)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    probs = []
    for i, (M, N) in enumerate(shapes):
        A = bf16_round(torch.randn(K, M, generator=g) * 0.5 + i)         # A^T stored [K][M]
        B = bf16_round(torch.randn(K, N, generator=g) * 0.5 - i)
        ref = (A.double().t() @ B.double())
        d = L.GemmDesc()
        d.dtype, d.trans_a, d.trans_b, d.M, d.N, d.K = 1, 1, 1, M, N, K
        d.lda, d.ldb, d.ldc, d.c_fp32, d.batch, d.batch_inner, d.alpha, d.accumulate = M, N, N, 1, 1, 1, 1.0, 1
        ws = torch.empty(8 * M * N, dtype=torch.float32, device=DEV) if slabs else None
        probs.append((d, to_dev_compute(A, 1), to_dev_compute(B, 1), torch.zeros(M, N, device=DEV), ws, ref))
    torch.cuda.synchronize()
    rounds = 12
    for r in range(rounds):
        for (d, A, B, Cd, ws, ref), st in zip(probs, streams):
            with torch.cuda.stream(st):
                Cd.zero_()
                L.check(l.b2s_gemm_splitk(C.byref(d), 8, A.data_ptr(), B.data_ptr(), Cd.data_ptr(), ws.data_ptr() if ws is not None else None,
                                          ws.numel() if ws is not None else 0, st.cuda_stream))
        torch.cuda.synchronize()
        for d, A, B, Cd, ws, ref in probs:
            assert relerr(Cd.cpu(), ref) < 2e-3, (r, report("splitk", Cd.cpu(), ref))


def test_gemm_tile_policy_switch():
    """b2s_gemm_set_tile_policy (include/b2s_hip.h): the data-parallel trainer's 256x128-everywhere policy against the per-shape choice on
    a projection-shaped GEMM (N = 768: 96-wide tiles by default) -- same product; a bad policy value is refused."""
    ops, lib = _ops()
    L = lib.load()
    from b2s_hip.lib import B2SError
    g = torch.Generator().manual_seed(11)
    M, N, K = 1300, 768, 192
    A, B = bf16_round(torch.randn(M, K, generator=g)), bf16_round(torch.randn(N, K, generator=g))
    ref = A.double() @ B.double().t()
    Ad, Bd = to_dev_compute(A, 1), to_dev_compute(B, 1)
    out = {}
    try:
        for pol in (0, 4, 3):
            assert L.b2s_gemm_set_tile_policy(pol) == 0
            out[pol] = ops.gemm(1, Ad, Bd, M, N, K, trans_a=False, trans_b=False, lda=K, ldb=K).cpu()      # (B stored [N, K]: the forward linear form)
            assert relerr(out[pol], ref) < TOL[1], pol
        assert L.b2s_gemm_set_tile_policy(5) != 0
        with pytest.raises(B2SError):
            lib.check(L.b2s_gemm_set_tile_policy(-1))
    finally:
        assert L.b2s_gemm_set_tile_policy(0) == 0
    assert torch.equal(out[0], out[3])              # (N = 768 picks the 96-wide tiles on its own)
