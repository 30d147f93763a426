"""Autograd for the HIP kernels: what lets `model.forward(...)` + `loss.backward()` of train.py:302-336 run the
encoder's matrix products, LayerNorms and the GLU + depthwise-conv core of the convolution module on
libfcenc_hip.so instead of ATen.

  * `linear(x, weight, bias)`            nn.Linear / 1x1 Conv1d (model.py:348-350, 376, 429, 520, 541, 579-582, 751).
    Forward and both backward products are the library's MFMA GEMM (out = A W^T):
        y  = x W^T + b            A = x  [M][K],   W = weight [N][K]                      `fcenc_gemm`
        dx = dy W                 A = dy [M][N],   W = weight^T [K][N]                    }  one call of
        dW = dy^T x               A = dy^T [N][M'], W = x^T [K][M']  (M' = M up to 64)    }  `fcenc_linear_bwd`, which
        db = column sums of dy                                                            }  also makes the transposes
    in exact fp32 (`v_mfma_f32_16x16x4_f32`) or, under autocast / compute dtype bf16, with bf16 operands and fp32
    accumulation; outputs and gradients are always fp32.
  * `layer_norm(x, weight, bias)`        nn.LayerNorm(d), eps 1e-5 (model.py:516, 578, 677): `fcenc_layernorm` forward,
    `fcenc_layernorm_bwd` backward (statistics recomputed from x, parameter gradients as deterministic partial sums).
  * `glu_dwconv(y, weight, bias)`        nn.GLU over channels + the causal depthwise convolution with zero history
    (model.py:521-525, 236-249) on token-major tensors, forward and backward in one pass each.

  * `bn_silu(z, bn)`                     nn.BatchNorm1d with batch statistics (running statistics updated in place)
    followed by nn.SiLU (model.py:528-537 in training mode) on token-major tensors: `fcenc_bn_silu_train_fwd` /
    `fcenc_bn_silu_train_bwd`.

The attention core (two batched products, shift, masks, softmax), dropout and the 3x3 stride-2 subsampling
convolutions stay on ATen in training; nothing here is used in eval mode, which runs the fused inference kernels.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _native as N

Tensor = torch.Tensor


def _gemm_f32out(a: Tensor, w: Tensor, bias: Optional[Tensor], dt: torch.dtype) -> Tensor:
    """a [M][K] @ w [N][K]^T (+ bias) -> fp32 [M][N]; operands cast to `dt` (fp32: exact, bf16: fp32 accumulate)."""
    m, k = a.shape
    n = w.shape[0]
    a = a.to(dt).contiguous()
    w = w.to(dt).contiguous()
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    N.check(N.lib().fcenc_gemm(N.dtype_code(dt), N.EPI_RESID, N.ptr(a), N.ptr(w), N.ptr(bias), None, N.ptr(out), None,
                               m, n, k, k, 0, 1.0, 0.0, N.stream_of(a)))
    return out


def gemm_shapes_ok(n: int, k: int) -> bool:
    """What `fcenc_gemm` takes in both directions: every matrix side that becomes a GEMM N a multiple of 128 and
    every K a multiple of 64 (all the encoder's Linear / pointwise layers at d_encoder % 128 == 0 do)."""
    return n % 128 == 0 and k % 128 == 0


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], bf16: bool):
        n, k = weight.shape
        dt = torch.bfloat16 if bf16 else torch.float32
        x2 = x.reshape(-1, k).float().contiguous()
        w2 = weight.float().contiguous()
        out = _gemm_f32out(x2, w2, None if bias is None else bias.float().contiguous(), dt)
        ctx.save_for_backward(x2, w2)
        ctx.dt, ctx.has_bias, ctx.in_shape = dt, bias is not None, x.shape
        return out.view(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, w2 = ctx.saved_tensors
        n, k = w2.shape
        dy2 = dy.reshape(-1, n).float().contiguous()
        m = dy2.shape[0]
        dev = dy2.device
        dx = torch.empty((m, k), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dw = torch.empty((n, k), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        db = torch.empty((n,), dtype=torch.float32, device=dev) if ctx.has_bias and ctx.needs_input_grad[2] else None
        code = N.dtype_code(ctx.dt)
        ws = torch.empty((N.lib().fcenc_linear_bwd_workspace_bytes(code, m, n, k),), dtype=torch.uint8, device=dev)
        N.check(N.lib().fcenc_linear_bwd(code, N.ptr(x2), N.ptr(dy2), N.ptr(w2), N.ptr(dx), N.ptr(dw), N.ptr(db), N.ptr(ws),
                                         m, n, k, N.stream_of(dy2)))
        return (None if dx is None else dx.view(ctx.in_shape)), dw, db, None


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, bf16: bool = False) -> Tensor:
    """x (..., K) @ weight (N, K)^T + bias on the HIP GEMM with autograd; fp32 in, fp32 out."""
    return _Linear.apply(x, weight, bias, bf16)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor):
        d = x.shape[-1]
        x2 = x.reshape(-1, d).float().contiguous()
        g, b = weight.float().contiguous(), bias.float().contiguous()
        y = torch.empty_like(x2)
        N.check(N.lib().fcenc_layernorm(N.F32, N.ptr(x2), N.ptr(g), N.ptr(b), None, None, N.ptr(y), None, None,
                                        x2.shape[0], d, 0, 0, N.stream_of(x2)))
        ctx.save_for_backward(x2, g)
        ctx.in_shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, g = ctx.saved_tensors
        m, d = x2.shape
        dy2 = dy.reshape(m, d).float().contiguous()
        dx = torch.empty_like(x2)
        nblocks = max(1, min(1024, (m + 3) // 4))
        partial = torch.empty((nblocks, 2, d), dtype=torch.float32, device=x2.device)
        N.check(N.lib().fcenc_layernorm_bwd(N.ptr(x2), N.ptr(dy2), N.ptr(g), N.ptr(dx), N.ptr(partial), nblocks, m, d,
                                            N.stream_of(x2)))
        sums = partial.sum(0)
        return dx.view(ctx.in_shape), sums[0], sums[1]


def layer_norm(x: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    """nn.LayerNorm over the last dimension (eps 1e-5) on the HIP kernels with autograd; d % 4 == 0, d <= 1024."""
    return _LayerNorm.apply(x, weight, bias)


class _GluDwConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y: Tensor, weight: Tensor, bias: Tensor):
        b, t, d2 = y.shape
        d = d2 // 2
        y = y.float().contiguous()
        w9 = weight.float().reshape(d, 9).t().contiguous()           # [9][d]
        bb = bias.float().contiguous()
        g = torch.empty((b, t, d), dtype=torch.float32, device=y.device)
        z = torch.empty_like(g)
        N.check(N.lib().fcenc_glu_dwconv_fwd(N.ptr(y), N.ptr(w9), N.ptr(bb), N.ptr(g), N.ptr(z), b, t, d, N.stream_of(y)))
        ctx.save_for_backward(y, g, w9)
        ctx.w_shape = weight.shape
        return z

    @staticmethod
    def backward(ctx, dz: Tensor):
        y, g, w9 = ctx.saved_tensors
        b, t, d = g.shape
        dz = dz.float().contiguous()
        dy = torch.empty_like(y)
        blocks = N.lib().fcenc_glu_dwconv_partial_blocks(b, t)
        partial = torch.empty((blocks, 10, d), dtype=torch.float32, device=y.device)
        N.check(N.lib().fcenc_glu_dwconv_bwd(N.ptr(y), N.ptr(g), N.ptr(dz), N.ptr(w9), N.ptr(dy), N.ptr(partial), b, t, d,
                                             N.stream_of(y)))
        sums = partial.sum(0)                                         # [10][d]
        return dy, sums[:9].t().reshape(ctx.w_shape), sums[9]


def glu_dwconv(y: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    """y (B, T, 2d) token-major pointwise1 output, weight (d, 1, 9) / bias (d,) of the depthwise Conv1d ->
    (B, T, d): GLU over channels, then the causal depthwise convolution with zero left padding."""
    if weight.shape[-1] != 9 or weight.shape[0] * 2 != y.shape[-1] or y.shape[-1] % 8:
        raise ValueError(f"glu_dwconv: y {tuple(y.shape)} / depthwise weight {tuple(weight.shape)} (kernel 9, d % 4 == 0)")
    return _GluDwConv.apply(y, weight, bias)


class _BnSilu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z: Tensor, weight: Tensor, bias: Tensor, running_mean, running_var, momentum: float, eps: float):
        d = z.shape[-1]
        z2 = z.reshape(-1, d).float().contiguous()
        m = z2.shape[0]
        g, b = weight.float().contiguous(), bias.float().contiguous()
        y = torch.empty_like(z2)
        stat = torch.empty((4, d), dtype=torch.float32, device=z2.device)
        part = torch.empty((N.lib().fcenc_bn_partial_blocks(m), 2, d), dtype=torch.float32, device=z2.device)
        N.check(N.lib().fcenc_bn_silu_train_fwd(N.ptr(z2), N.ptr(g), N.ptr(b), N.ptr(running_mean), N.ptr(running_var),
                                                momentum, eps, N.ptr(y), N.ptr(stat), N.ptr(part), m, d, N.stream_of(z2)))
        ctx.save_for_backward(z2, g, b, stat)
        ctx.in_shape = z.shape
        return y.view(z.shape)

    @staticmethod
    def backward(ctx, dy: Tensor):
        z2, g, b, stat = ctx.saved_tensors
        m, d = z2.shape
        dy2 = dy.reshape(m, d).float().contiguous()
        dz = torch.empty_like(z2)
        dg = torch.empty((d,), dtype=torch.float32, device=z2.device)
        db = torch.empty_like(dg)
        part = torch.empty((N.lib().fcenc_bn_partial_blocks(m), 2, d), dtype=torch.float32, device=z2.device)
        N.check(N.lib().fcenc_bn_silu_train_bwd(N.ptr(z2), N.ptr(dy2), N.ptr(g), N.ptr(b), N.ptr(stat), N.ptr(dz), N.ptr(dg),
                                                N.ptr(db), N.ptr(part), m, d, N.stream_of(z2)))
        return dz.view(ctx.in_shape), dg, db, None, None, None, None


def bn_silu_ok(bn: torch.nn.Module, d: int) -> bool:
    """What `bn_silu` takes: an affine nn.BatchNorm1d in training mode with fp32 running statistics (or none) and a
    fixed momentum, d % 4 == 0, d <= 1024."""
    return (isinstance(bn, torch.nn.BatchNorm1d) and bn.training and bn.affine and bn.momentum is not None
            and d % 4 == 0 and d <= 1024
            and (not bn.track_running_stats or (bn.running_mean.dtype == torch.float32 and bn.running_mean.is_contiguous()
                                                and bn.running_var.is_contiguous())))


def bn_silu(z: Tensor, bn: torch.nn.BatchNorm1d) -> Tensor:
    """silu(bn(z)) for token-major z (..., d) in training mode: statistics over all leading dimensions, running
    statistics and num_batches_tracked updated as nn.BatchNorm1d does (no_grad, in place)."""
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    y = _BnSilu.apply(z, bn.weight, bn.bias, rm, rv, float(bn.momentum), float(bn.eps))
    if bn.track_running_stats:
        # the kernel wrote the running statistics through raw pointers: tell autograd's version counters, which is
        # what the encoder's packed-weight cache watches
        torch.autograd.graph.increment_version(rm)
        torch.autograd.graph.increment_version(rv)
        if bn.num_batches_tracked is not None:
            with torch.no_grad():
                bn.num_batches_tracked.add_(1)
    return y
