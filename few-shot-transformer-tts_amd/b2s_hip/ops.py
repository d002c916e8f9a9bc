"""Op-level autograd wrappers over the C ABI (b2s_gemm, b2s_layernorm_*, b2s_attention_*).

Used by the standalone modules (MultiheadAttention, FFNLayer, DecoderPrenet called on their own) and by
the op parity tests.  Tensors cross this layer as fp32; in bf16 mode the operands are cast to bf16 on
the device by b2s_cast before the MFMA kernels run.
"""
import ctypes as C

import torch

from . import lib as L


def _rup8(x):
    return (x + 7) // 8 * 8


def _esz(dtype):
    return 2 if dtype else 4


def to_compute(x, dtype):
    """fp32 tensor -> compute-dtype device buffer (bf16 raw bits as int16 tensor, or the tensor itself)."""
    x = x.contiguous()
    if not dtype:
        return x
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    L.check(L.load().b2s_cast(1, L.ptr(x), L.ptr(out), x.numel(), L.stream()))
    return out


def from_compute(x, dtype):
    if not dtype:
        return x
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    L.check(L.load().b2s_cast_back(1, L.ptr(x), L.ptr(out), x.numel(), L.stream()))
    return out


def gemm(dtype, A, B, M, N, K, trans_a=False, trans_b=False, out=None, c_fp32=True, lda=None, ldb=None, ldc=None,
         bias=None, relu=False, residual=None, accumulate=False, alpha=1.0, batch=1, batch_inner=1,
         a_bs=(0, 0), b_bs=(0, 0), c_bs=(0, 0), row_len=None, rows_per_batch=1, conv_cin_a=0, conv_T=0, conv_len=None,
         conv_dw_cin=0, drop_p=0.0, seed=0):
    """Raw b2s_gemm call; A/B are compute-dtype device tensors."""
    d = L.GemmDesc()
    d.dtype, d.trans_a, d.trans_b, d.M, d.N, d.K = dtype, int(trans_a), int(trans_b), M, N, K
    d.lda = lda if lda is not None else (M if trans_a else K)
    d.ldb = ldb if ldb is not None else (N if trans_b else K)
    d.ldc = ldc if ldc is not None else N
    d.c_fp32 = int(c_fp32)
    d.batch, d.batch_inner = batch, batch_inner
    d.a_bs_o, d.a_bs_i = a_bs
    d.b_bs_o, d.b_bs_i = b_bs
    d.c_bs_o, d.c_bs_i = c_bs
    d.alpha, d.relu, d.accumulate, d.drop_p, d.seed = alpha, int(relu), int(accumulate), drop_p, seed
    d.conv_cin_a, d.conv_T, d.conv_dw_cin, d.rows_per_batch = conv_cin_a, conv_T, conv_dw_cin, rows_per_batch
    if out is None:
        shape = (batch, M, d.ldc) if batch > 1 else (M, d.ldc)
        out = torch.empty(shape, dtype=torch.float32 if c_fp32 or not dtype else torch.int16, device=A.device)
    L.check(L.load().b2s_gemm(C.byref(d), L.ptr(A), L.ptr(B), L.ptr(out), L.ptr(bias), L.ptr(residual), L.ptr(row_len),
                              L.ptr(conv_len), L.stream()))
    return out


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ b) (optionally ReLU): nn.Linear of the reference (attention.py:43-47, modules.py:11-13)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, dtype):
        K = x.shape[-1]
        N = weight.shape[0]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        xT, wT = to_compute(x2, dtype), to_compute(weight, dtype)
        y = gemm(dtype, xT, wT, M, N, K, bias=bias.contiguous() if bias is not None else None, relu=relu)
        ctx.save_for_backward(xT, wT, y if relu else None)
        ctx.dtype, ctx.shape, ctx.has_bias, ctx.relu = dtype, x.shape, bias is not None, relu
        return y.reshape(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        xT, wT, y = ctx.saved_tensors
        dtype = ctx.dtype
        M, K = xT.shape
        N = wT.shape[0]
        dy2 = dy.reshape(M, N)
        if ctx.relu:
            dy2 = dy2 * (y > 0)          # tiny standalone-module path only
        dyT = to_compute(dy2, dtype)
        dx = gemm(dtype, dyT, wT, M, K, N, trans_b=True)                       # dX = dY W
        dw = gemm(dtype, dyT, xT, N, K, M, trans_a=True, trans_b=True)         # dW = dY^T X
        db = None
        if ctx.has_bias:
            ones = to_compute(torch.ones(M, 8, device=dy.device), dtype)
            db = gemm(dtype, dyT, ones, N, 8, M, trans_a=True, trans_b=True)[:, 0].contiguous()
        return dx.reshape(ctx.shape), dw, db, None, None


def linear(x, weight, bias=None, relu=False, dtype=0):
    return LinearFn.apply(x, weight, bias, relu, dtype)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, dtype):
        D = x.shape[-1]
        x2 = x.reshape(-1, D).contiguous()
        M = x2.shape[0]
        y = torch.empty(M, D, dtype=torch.int16 if dtype else torch.float32, device=x.device)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        L.check(L.load().b2s_layernorm_forward(dtype, L.ptr(x2), L.ptr(weight), L.ptr(bias), L.ptr(y), L.ptr(mean),
                                               L.ptr(rstd), M, D, eps, L.stream()))
        ctx.save_for_backward(x2, weight, mean, rstd)
        ctx.dtype, ctx.shape = dtype, x.shape
        return from_compute(y, dtype).reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, mean, rstd = ctx.saved_tensors
        M, D = x2.shape
        dyT = to_compute(dy.reshape(M, D), ctx.dtype)
        dx = torch.empty_like(x2)
        dg = torch.empty(D, dtype=torch.float32, device=x2.device)
        db = torch.empty_like(dg)
        L.check(L.load().b2s_layernorm_backward(ctx.dtype, L.ptr(dyT), L.ptr(x2), L.ptr(weight), L.ptr(mean), L.ptr(rstd),
                                                L.ptr(dx), L.ptr(dg), L.ptr(db), M, D, L.stream()))
        return dx.reshape(ctx.shape), dg, db, None, None


def dropout_mask(shape, p, seed, op_id, device):
    """fp32 tensor of 0 / 1/(1-p) drawn from the engine's counter RNG (b2s_dropout_mask: the generator the kernels use)."""
    n = 1
    for d in shape:
        n *= int(d)
    m = torch.empty(n, dtype=torch.uint8, device=device)
    L.check(L.load().b2s_dropout_mask(float(p), int(seed), int(op_id), L.ptr(m), n, L.stream()))
    return m.reshape(shape).to(torch.float32) * (1.0 / (1.0 - p))


def layernorm(x, weight, bias, eps=1e-6, dtype=0):
    return LayerNormFn.apply(x, weight, bias, eps, dtype)


class AttentionCoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh) + mask) v on [B, L, H*dh] tensors; returns (context, probs [B,H,Lq,Lk])."""

    @staticmethod
    def forward(ctx, q, k, v, H, mask_mode, klen, bias, drop_p, seed, dtype):
        B, Lq, Cq = q.shape
        Lk = k.shape[1]
        dh = Cq // H
        lib = L.load()
        qT, kT, vT = to_compute(q, dtype), to_compute(k, dtype), to_compute(v, dtype)
        ldp = _rup8(Lk)
        ws = torch.empty(lib.b2s_attention_ws_bytes(dtype, B, H, Lq, Lk), dtype=torch.uint8, device=q.device)
        P = torch.empty(B, H, Lq, ldp, dtype=torch.int16 if dtype else torch.float32, device=q.device)
        Pd = torch.empty_like(P) if drop_p > 0 else None
        out = torch.empty(B, Lq, Cq, dtype=P.dtype, device=q.device)
        bias_sb = bias_sq = 0
        if bias is not None:
            bias = bias.to(torch.float32).expand(bias.shape[0], 1, bias.shape[2], Lk).contiguous()
            bias_sb = bias.shape[2] * Lk if bias.shape[0] > 1 else 0
            bias_sq = Lk if bias.shape[2] > 1 else 0
        L.check(lib.b2s_attention_forward(dtype, L.ptr(qT), Cq, L.ptr(kT), Cq, L.ptr(vT), Cq, L.ptr(out), Cq, B, H, Lq, Lk, dh,
                                          mask_mode, L.ptr(klen), L.ptr(bias), bias_sb, bias_sq, drop_p, seed, L.ptr(ws),
                                          L.ptr(P), L.ptr(Pd), L.stream()))
        ctx.save_for_backward(qT, kT, vT, P, Pd)
        ctx.cfg = (B, H, Lq, Lk, dh, drop_p, seed, dtype)
        probs = from_compute(P, dtype)[..., :Lk]
        ctx.mark_non_differentiable(probs)
        return from_compute(out, dtype), probs

    @staticmethod
    def backward(ctx, dctx, _dprobs):
        qT, kT, vT, P, Pd = ctx.saved_tensors
        B, H, Lq, Lk, dh, drop_p, seed, dtype = ctx.cfg
        lib = L.load()
        Cq = H * dh
        dT = to_compute(dctx, dtype)
        ws = torch.empty(lib.b2s_attention_ws_bytes(dtype, B, H, Lq, Lk), dtype=torch.uint8, device=dctx.device)
        dq, dk, dv = torch.empty_like(qT), torch.empty_like(kT), torch.empty_like(vT)
        L.check(lib.b2s_attention_backward(dtype, L.ptr(dT), Cq, L.ptr(qT), Cq, L.ptr(kT), Cq, L.ptr(vT), Cq, L.ptr(P), L.ptr(Pd),
                                           L.ptr(dq), Cq, L.ptr(dk), Cq, L.ptr(dv), Cq, B, H, Lq, Lk, dh, drop_p, seed, L.ptr(ws),
                                           L.stream()))
        return (from_compute(dq, dtype), from_compute(dk, dtype), from_compute(dv, dtype)) + (None,) * 7


class FlashAttentionFn(torch.autograd.Function):
    """Fused attention core (csrc/attention.hip): same contract as AttentionCoreFn, logits never materialised; the
    returned weights are recomputed on demand from the saved log-sum-exp."""

    @staticmethod
    def forward(ctx, q, k, v, H, mask_mode, klen, drop_p, seed, dtype, want_probs):
        B, Lq, Cq = q.shape
        Lk = k.shape[1]
        dh = Cq // H
        lib = L.load()
        qT, kT, vT = to_compute(q, dtype), to_compute(k, dtype), to_compute(v, dtype)
        out = torch.empty(B, Lq, Cq, dtype=qT.dtype, device=q.device)
        lse = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
        L.check(lib.b2s_flash_attention_forward(dtype, L.ptr(qT), Cq, L.ptr(kT), Cq, L.ptr(vT), Cq, L.ptr(out), Cq, B, H, Lq, Lk, dh,
                                                mask_mode, L.ptr(klen), drop_p, seed, L.ptr(lse), L.stream()))
        probs = torch.empty(0, device=q.device)
        if want_probs:
            al = torch.empty(B, H, Lk, Lq, dtype=torch.float32, device=q.device)
            L.check(lib.b2s_flash_attention_align(dtype, L.ptr(qT), Cq, L.ptr(kT), Cq, L.ptr(lse), B, H, Lq, Lk, dh, mask_mode,
                                                  L.ptr(klen), L.ptr(al), L.stream()))
            probs = al.permute(0, 1, 3, 2)
        ctx.save_for_backward(qT, kT, vT, out, lse, klen)
        ctx.cfg = (B, H, Lq, Lk, dh, mask_mode, drop_p, seed, dtype)
        ctx.mark_non_differentiable(probs)
        return from_compute(out, dtype), probs

    @staticmethod
    def backward(ctx, dctx, _dprobs):
        qT, kT, vT, out, lse, klen = ctx.saved_tensors
        B, H, Lq, Lk, dh, mask_mode, drop_p, seed, dtype = ctx.cfg
        lib = L.load()
        Cq = H * dh
        dT = to_compute(dctx, dtype)
        dsum = torch.empty(B, H, Lq, dtype=torch.float32, device=dctx.device)
        dq, dk, dv = torch.empty_like(qT), torch.empty_like(kT), torch.empty_like(vT)
        L.check(lib.b2s_flash_attention_backward(dtype, L.ptr(dT), L.ptr(out), Cq, L.ptr(qT), Cq, L.ptr(kT), Cq, L.ptr(vT), Cq, L.ptr(lse),
                                                 L.ptr(dsum), L.ptr(dq), Cq, L.ptr(dk), Cq, L.ptr(dv), Cq, B, H, Lq, Lk, dh, mask_mode,
                                                 L.ptr(klen), drop_p, seed, L.stream()))
        return (from_compute(dq, dtype), from_compute(dk, dtype), from_compute(dv, dtype)) + (None,) * 7


def attention_core(q, k, v, H, mask_mode=0, klen=None, bias=None, drop_p=0.0, seed=0, dtype=0, fused=None):
    """fused=None: use the fused kernel whenever it applies (head size 32/64/96 and no dense bias)."""
    dh = q.shape[-1] // H
    can = bias is None and dh in (32, 64, 96)
    if fused is None:
        fused = can
    if fused:
        if not can:
            raise ValueError("fused attention needs head size 32/64/96 and no dense bias")
        return FlashAttentionFn.apply(q, k, v, H, mask_mode, klen, drop_p, seed, dtype, True)
    return AttentionCoreFn.apply(q, k, v, H, mask_mode, klen, bias, drop_p, seed, dtype)
