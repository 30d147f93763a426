"""Tensor-level wrappers of the libfcenc_hip.so kernels, registered with torch.library
(namespace `fcenc`).  Argument validation (dtype, contiguity, shape) happens here, in the
style of the reference's own asserts (model.py:285), before the raw pointers cross the C ABI.
Every op runs on the tensor's current HIP stream and raises RuntimeError with the library's
message on failure; none has a non-HIP implementation.
"""
from __future__ import annotations

import ctypes as C
import itertools
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _native as N

Tensor = torch.Tensor


def _cdt(t: Tensor) -> int:
    return N.dtype_code(t.dtype)


# ------------------------------------------------------------------------------------------
# whole passes: what FastConformerEncoder.forward dispatches to (model.py:835-894)
# ------------------------------------------------------------------------------------------
# An operator's arguments are tensors and scalars; the packed weights of an encoder (a few hundred device
# tensors plus their C descriptor) and a stream's ring buffers are Python objects.  They are passed as integer
# handles into this table of weak references: the owner (the module's `_packed` cache, the caller's `new_cache`)
# keeps them alive, a handle whose owner is gone raises.
_HANDLES: Dict[int, "weakref.ReferenceType"] = {}
_next_handle = itertools.count(1)


def register_handle(obj) -> int:
    h = next(_next_handle)
    _HANDLES[h] = weakref.ref(obj, lambda _r, h=h: _HANDLES.pop(h, None))
    return h


def from_handle(h: int):
    ref = _HANDLES.get(h)
    obj = ref() if ref is not None else None
    if obj is None:
        raise RuntimeError(f"fcenc: handle {h} does not name a live packed encoder / stream state")
    return obj


def _out_frames(t: int) -> int:
    for _ in range(3):
        t = (t + 1) // 2
    return t


@torch.library.custom_op("fcenc::encoder_forward", mutates_args=())
def encoder_forward(mel: Tensor, lengths: Optional[Tensor], handle: int, d_model: int, lanes: int
                    ) -> Tuple[Tensor, Tensor]:
    """The offline encoder pass as ONE operator (one C call, `fcenc_encoder_forward_lanes`): mel (B, T, n_mels) fp32
    positive power-mel, lengths (B,) frames before subsampling or None -> (out (B, T', d_model) fp32,
    output_lengths (B,) int64).  `handle` names the PackedEncoder (weights, compute dtype, workspace)."""
    pk = from_handle(handle)
    out, out_len, _mask = pk.forward(mel, lengths, pk.workspace, lanes=lanes)
    return out, out_len


@encoder_forward.register_fake
def _(mel, lengths, handle, d_model, lanes):
    b, t, _ = mel.shape
    return (mel.new_empty((b, _out_frames(t), d_model), dtype=torch.float32),
            mel.new_empty((b,), dtype=torch.int64))


@torch.library.custom_op("fcenc::encoder_stream_step", mutates_args=("kcache", "vcache", "conv_cache"))
def encoder_stream_step(mel: Tensor, kcache: Sequence[Tensor], vcache: Sequence[Tensor], conv_cache: Sequence[Tensor],
                        handle: int, state_handle: int, d_model: int, lanes: int, graph: bool) -> Tensor:
    """One chunk of every stream as ONE operator (`fcenc_encoder_stream_step`): mel (B, T, n_mels) fp32 ->
    (B, T', d_model) fp32.  The per-layer K / V rings and depthwise-conv frames of the StreamCache named by
    `state_handle` are listed as arguments because the step rewrites them in place."""
    pk, state = from_handle(handle), from_handle(state_handle)
    return pk.stream_step(mel, state, pk.stream_workspace, lanes, graph)


@encoder_stream_step.register_fake
def _(mel, kcache, vcache, conv_cache, handle, state_handle, d_model, lanes, graph):
    b, t, _ = mel.shape
    return mel.new_empty((b, _out_frames(t), d_model), dtype=torch.float32)


@torch.library.custom_op("fcenc::mel_spectrogram", mutates_args=())
def mel_spectrogram(wave: Tensor, lengths: Optional[Tensor], dft: Tensor, win: Tensor, fb_start: Tensor,
                    fb_count: Tensor, fb_w: Tensor, n_fft: int, hop: int, kp: int, wpitch: int, n_mels: int,
                    fb_maxw: int) -> Tuple[Tensor, Tensor]:
    """Padded waveforms (B, S) fp32 (+ sample counts) -> ((B, 1 + S // hop, n_mels) fp32 power-mel with zeros past each
    utterance's frames, (B,) int64 frame counts): one launch of `fcenc_mel_spectrogram`; the plan tensors are
    those of conformer_amd.frontend.build_plan."""
    if not wave.is_cuda:
        raise RuntimeError("fcenc: the mel front-end runs on MI355X through libfcenc_hip.so only")
    if wave.dim() != 2 or wave.dtype != torch.float32 or wave.stride(1) != 1:
        raise ValueError("fcenc::mel_spectrogram: wave must be (batch, samples) fp32 with unit sample stride")
    b, s = wave.shape
    if lengths is not None:
        N.require(lengths, torch.int64, "lengths", 1)
        if lengths.numel() != b:
            raise ValueError("fcenc::mel_spectrogram: lengths must hold one sample count per utterance")
    t = 1 + s // hop
    mel = torch.empty((b, t, n_mels), dtype=torch.float32, device=wave.device)
    frames = torch.empty((b,), dtype=torch.int64, device=wave.device)
    plan = N.MelPlan(n_fft, hop, kp, wpitch, n_mels, fb_maxw, N.ptr(dft), N.ptr(win), N.ptr(fb_start),
                     N.ptr(fb_count), N.ptr(fb_w))
    N.check(N.lib().fcenc_mel_spectrogram(C.byref(plan), N.ptr(wave), wave.stride(0), N.ptr(lengths), b, s,
                                          N.ptr(mel), t, N.ptr(frames), N.stream_of(wave)))
    return mel, frames


@mel_spectrogram.register_fake
def _(wave, lengths, dft, win, fb_start, fb_count, fb_w, n_fft, hop, kp, wpitch, n_mels, fb_maxw):
    b, s = wave.shape
    return (wave.new_empty((b, 1 + s // hop, n_mels), dtype=torch.float32),
            wave.new_empty((b,), dtype=torch.int64))


# ------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------
@torch.library.custom_op("fcenc::linear", mutates_args=())
def linear(a: Tensor, w: Tensor, bias: Optional[Tensor], act: int) -> Tensor:
    """act 0: a @ w.T + bias; 1: silu(...); 2: GLU over 16/16-interleaved weight rows."""
    N.require(a, a.dtype, "a", 2)
    N.require(w, a.dtype, "w", 2)
    if bias is not None:
        N.require(bias, torch.float32, "bias", 1)
    m, k = a.shape
    n = w.shape[0]
    if w.shape[1] != k:
        raise ValueError(f"fcenc::linear: a is {tuple(a.shape)} but w is {tuple(w.shape)}")
    cols = n // 2 if act == N.EPI_GLU else n
    out = torch.empty((m, cols), dtype=a.dtype, device=a.device)
    N.check(N.lib().fcenc_gemm(_cdt(a), act, N.ptr(a), N.ptr(w), N.ptr(bias), N.ptr(out), None, None,
                               m, n, k, k, cols, 1.0, 0.0, N.stream_of(a)))
    return out


@linear.register_fake
def _(a, w, bias, act):
    n = w.shape[0]
    return a.new_empty((a.shape[0], n // 2 if act == N.EPI_GLU else n))


@torch.library.custom_op("fcenc::linear_residual_", mutates_args=("x", "xc"))
def linear_residual_(x: Tensor, a: Tensor, w: Tensor, bias: Optional[Tensor], alpha: float, beta: float,
                     xc: Optional[Tensor]) -> None:
    """x = beta*x + alpha*(a @ w.T + bias) on the fp32 residual stream; xc (optional) receives x in a's dtype."""
    N.require(x, torch.float32, "x", 2)
    N.require(a, a.dtype, "a", 2)
    N.require(w, a.dtype, "w", 2)
    if bias is not None:
        N.require(bias, torch.float32, "bias", 1)
    if xc is not None:
        N.require(xc, a.dtype, "xc", 2)
        if xc.shape != x.shape:
            raise ValueError("fcenc::linear_residual_: xc must have x's shape")
    m, k = a.shape
    n = w.shape[0]
    if w.shape[1] != k or tuple(x.shape) != (m, n):
        raise ValueError("fcenc::linear_residual_: shape mismatch")
    N.check(N.lib().fcenc_gemm(_cdt(a), N.EPI_RESID, N.ptr(a), N.ptr(w), N.ptr(bias), None, N.ptr(x), N.ptr(xc),
                               m, n, k, k, 0, alpha, beta, N.stream_of(a)))


@torch.library.custom_op("fcenc::conv3x3s2_bn_silu", mutates_args=())
def conv3x3s2_bn_silu(x: Tensor, w: Tensor, bias: Tensor, hout_pad: int) -> Tensor:
    """x [B][H][W][Cin] channels-last, w [Cout][9*Cin] (BatchNorm folded) -> [B][hout_pad][ceil(W/2)][Cout];
    rows >= ceil(H/2) are zero."""
    N.require(x, x.dtype, "x", 4)
    N.require(w, x.dtype, "w", 2)
    N.require(bias, torch.float32, "bias", 1)
    b, h, wd, cin = x.shape
    cout = w.shape[0]
    if w.shape[1] != 9 * cin:
        raise ValueError("fcenc::conv3x3s2_bn_silu: weight must be [Cout][9*Cin]")
    hout = (h + 1) // 2
    hp = max(hout_pad, hout)
    out = torch.zeros((b, hp, (wd + 1) // 2, cout), dtype=x.dtype, device=x.device) if hp > hout else \
        torch.empty((b, hp, (wd + 1) // 2, cout), dtype=x.dtype, device=x.device)
    N.check(N.lib().fcenc_conv3x3s2(_cdt(x), N.ptr(x), N.ptr(w), N.ptr(bias), N.ptr(out), b, h, wd, cin, cout, hp,
                                    N.stream_of(x)))
    return out


@conv3x3s2_bn_silu.register_fake
def _(x, w, bias, hout_pad):
    b, h, wd, _ = x.shape
    return x.new_empty((b, max(hout_pad, (h + 1) // 2), (wd + 1) // 2, w.shape[0]))


@torch.library.custom_op("fcenc::conv3_patch", mutates_args=())
def conv3_patch(x: Tensor, w: Tensor, bias: Tensor, hout_pad: int) -> Tensor:
    """conv3x3s2_bn_silu for the third subsampling convolution's shape (bf16, 256 channels, 20 input columns), with a
    tile's input patch staged once in LDS.  x [B][H][20][256], w [256][9*256] (k = tap*256 + channel, BatchNorm
    folded) -> [B][hout_pad][10][256]; rows >= ceil(H/2) are zero."""
    N.require(x, torch.bfloat16, "x", 4)
    N.require(w, torch.bfloat16, "w", 2)
    N.require(bias, torch.float32, "bias", 1)
    b, h, wd, c = x.shape
    lib = N.lib()
    if not lib.fcenc_conv3_patch_supported(N.BF16, wd, c, w.shape[0]) or w.shape[1] != 9 * c:
        raise ValueError(f"fcenc: conv3_patch takes 20 input columns and 256 channels, got {tuple(x.shape)} / {tuple(w.shape)}")
    hout = (h + 1) // 2
    hp = max(hout_pad, hout)
    out = (torch.zeros if hp > hout else torch.empty)((b, hp, (wd + 1) // 2, c), dtype=x.dtype, device=x.device)
    wp = torch.empty(lib.fcenc_conv3_patch_packed_bytes(), dtype=torch.uint8, device=x.device)
    s = N.stream_of(x)
    N.check(lib.fcenc_conv3_patch_pack(N.ptr(w), N.ptr(wp), c, s))
    N.check(lib.fcenc_conv3_patch(N.ptr(x), N.ptr(wp), N.ptr(bias), N.ptr(out), b, h, wd, c, hp, s))
    return out


@conv3_patch.register_fake
def _(x, w, bias, hout_pad):
    b, h, wd, _ = x.shape
    return x.new_empty((b, max(hout_pad, (h + 1) // 2), (wd + 1) // 2, w.shape[0]))


# ------------------------------------------------------------------------------------------
# memory-bound kernels
# ------------------------------------------------------------------------------------------
@torch.library.custom_op("fcenc::lognorm_conv1", mutates_args=())
def lognorm_conv1(mel: Tensor, w9: Tensor, bias: Tensor, bf16_out: bool) -> Tensor:
    """log + per-utterance standardisation + Conv2d(1->C,3x3,s2,p1) + folded BatchNorm + SiLU -> [B][T1][F1][C]."""
    N.require(mel, torch.float32, "mel", 3)
    N.require(w9, torch.float32, "w9", 2)
    N.require(bias, torch.float32, "bias", 1)
    b, t, f = mel.shape
    c = w9.shape[1]
    nsplit = 16
    partial = torch.empty((b, nsplit, 2), dtype=torch.float64, device=mel.device)
    s = N.stream_of(mel)
    N.check(N.lib().fcenc_input_stats(N.ptr(mel), N.ptr(partial), b, t, f, nsplit, s))
    dt = torch.bfloat16 if bf16_out else torch.float32
    out = torch.empty((b, (t + 1) // 2, (f + 1) // 2, c), dtype=dt, device=mel.device)
    N.check(N.lib().fcenc_conv1(N.dtype_code(dt), N.ptr(mel), N.ptr(partial), nsplit, N.ptr(w9), N.ptr(bias),
                                N.ptr(out), b, t, f, c, s))
    return out


@lognorm_conv1.register_fake
def _(mel, w9, bias, bf16_out):
    b, t, f = mel.shape
    return mel.new_empty((b, (t + 1) // 2, (f + 1) // 2, w9.shape[1]),
                         dtype=torch.bfloat16 if bf16_out else torch.float32)


@torch.library.custom_op("fcenc::lognorm_conv12", mutates_args=())
def lognorm_conv12(mel: Tensor, w9: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """log + per-utterance standardisation + the first two subsampling convolutions (each with folded BatchNorm and
    SiLU) in one kernel; conv1's activation stays on chip.  mel [B][T][80] fp32; w9 [9][256], b1 [256] fp32 (conv1);
    w2 [256][9*256] bf16 (k = tap*256 + channel), b2 [256] fp32 (conv2) -> [B][ceil(ceil(T/2)/2)][20][256] bf16."""
    N.require(mel, torch.float32, "mel", 3)
    N.require(w9, torch.float32, "w9", 2)
    N.require(b1, torch.float32, "b1", 1)
    N.require(w2, torch.bfloat16, "w2", 2)
    N.require(b2, torch.float32, "b2", 1)
    b, t, f = mel.shape
    c = w9.shape[1]
    lib = N.lib()
    if not lib.fcenc_conv12_supported(N.BF16, f, c) or tuple(w2.shape) != (c, 9 * c) or w9.shape[0] != 9:
        raise ValueError(f"fcenc: lognorm_conv12 takes 80 mel bins and 256 channels, got {f} / {tuple(w9.shape)} / {tuple(w2.shape)}")
    nsplit = 16
    partial = torch.empty((b, nsplit, 2), dtype=torch.float64, device=mel.device)
    a1p = torch.empty(lib.fcenc_conv12_packed_bytes(0), dtype=torch.uint8, device=mel.device)
    w2p = torch.empty(lib.fcenc_conv12_packed_bytes(1), dtype=torch.uint8, device=mel.device)
    s = N.stream_of(mel)
    N.check(lib.fcenc_conv12_pack(N.ptr(w9), N.ptr(b1), N.ptr(w2), N.ptr(a1p), N.ptr(w2p), f, c, s))
    N.check(lib.fcenc_input_stats(N.ptr(mel), N.ptr(partial), b, t, f, nsplit, s))
    t2 = ((t + 1) // 2 + 1) // 2
    out = torch.empty((b, t2, 20, c), dtype=torch.bfloat16, device=mel.device)
    N.check(lib.fcenc_lognorm_conv12(N.ptr(mel), N.ptr(partial), nsplit, N.ptr(a1p), N.ptr(w2p), N.ptr(b2), N.ptr(out),
                                     b, t, f, c, s))
    return out


@lognorm_conv12.register_fake
def _(mel, w9, b1, w2, b2):
    b, t, _ = mel.shape
    return mel.new_empty((b, ((t + 1) // 2 + 1) // 2, 20, w9.shape[1]), dtype=torch.bfloat16)


@torch.library.custom_op("fcenc::layernorm", mutates_args=())
def layernorm(x: Tensor, g: Tensor, b: Tensor, g2: Optional[Tensor], b2: Optional[Tensor], bf16_out: bool
              ) -> Tuple[Tensor, Tensor]:
    """y1 = LN(x; g, b) (fp32) and y2 = LN(y1; g2, b2) if given else y1, in the compute dtype."""
    N.require(x, torch.float32, "x", 2)
    m, d = x.shape
    for nm, t in (("g", g), ("b", b), ("g2", g2), ("b2", b2)):
        if t is not None:
            N.require(t, torch.float32, nm, 1)
    dt = torch.bfloat16 if bf16_out else torch.float32
    y1 = torch.empty_like(x)
    y2 = torch.empty((m, d), dtype=dt, device=x.device)
    N.check(N.lib().fcenc_layernorm(N.dtype_code(dt), N.ptr(x), N.ptr(g), N.ptr(b), N.ptr(g2), N.ptr(b2), N.ptr(y1),
                                    N.ptr(y2), None, m, d, 0, 0, N.stream_of(x)))
    return y1, y2


@layernorm.register_fake
def _(x, g, b, g2, b2, bf16_out):
    return torch.empty_like(x), x.new_empty(x.shape, dtype=torch.bfloat16 if bf16_out else torch.float32)


@torch.library.custom_op("fcenc::dwconv9_bn_silu", mutates_args=())
def dwconv9_bn_silu(x: Tensor, w9: Tensor, bias: Tensor) -> Tensor:
    """x [B][T][d]; causal depthwise conv k=9 along T (zero history) + folded BatchNorm + SiLU."""
    N.require(x, x.dtype, "x", 3)
    N.require(w9, torch.float32, "w9", 2)
    N.require(bias, torch.float32, "bias", 1)
    b, t, d = x.shape
    if tuple(w9.shape) != (9, d):
        raise ValueError("fcenc::dwconv9_bn_silu: w9 must be [9][d]")
    out = torch.empty_like(x)
    N.check(N.lib().fcenc_dwconv(_cdt(x), N.ptr(x), N.ptr(w9), N.ptr(bias), N.ptr(out), b, t, d, N.stream_of(x)))
    return out


@dwconv9_bn_silu.register_fake
def _(x, w9, bias):
    return torch.empty_like(x)


@torch.library.custom_op("fcenc::relpos_attention", mutates_args=())
def relpos_attention(qkv: Tensor, pos: Tensor, bias_u: Tensor, bias_v: Tensor, klen: Tensor, batch: int,
                     t_valid: int, window: int, scale: float) -> Tensor:
    """qkv [B*Tp][3*H*64], pos [2*Tp+64][H*64], klen int32 [B] -> [B*Tp][H*64]."""
    N.require(qkv, qkv.dtype, "qkv", 2)
    N.require(pos, qkv.dtype, "pos", 2)
    N.require(bias_u, torch.float32, "bias_u", 1)
    N.require(bias_v, torch.float32, "bias_v", 1)
    N.require(klen, torch.int32, "klen", 1)
    hd = qkv.shape[1] // 3
    tp = qkv.shape[0] // batch
    if qkv.shape[0] != batch * tp or hd % 64 or tp % 64:
        raise ValueError("fcenc::relpos_attention: qkv must be [B*Tp][3*H*64] with Tp % 64 == 0")
    if tuple(pos.shape) != (2 * tp + 64, hd) or bias_u.numel() != hd or bias_v.numel() != hd or klen.numel() != batch:
        raise ValueError("fcenc::relpos_attention: operand shapes do not match qkv")
    out = torch.empty((batch * tp, hd), dtype=qkv.dtype, device=qkv.device)
    N.check(N.lib().fcenc_attention(_cdt(qkv), N.ptr(qkv), N.ptr(pos), N.ptr(bias_u), N.ptr(bias_v), N.ptr(klen),
                                    N.ptr(out), batch, hd // 64, tp, t_valid, window, scale, N.stream_of(qkv)))
    return out


@relpos_attention.register_fake
def _(qkv, pos, bias_u, bias_v, klen, batch, t_valid, window, scale):
    return qkv.new_empty((qkv.shape[0], qkv.shape[1] // 3))


@torch.library.custom_op("fcenc::output_lengths", mutates_args=())
def output_lengths(lengths: Optional[Tensor], batch: int, t_valid: int, device: torch.device
                   ) -> Tuple[Tensor, Tensor, Tensor]:
    """(out_len int64 [B], mask bool [B][T'] (all False when lengths is None), klen int32 [B])."""
    if lengths is not None:
        N.require(lengths, torch.int64, "lengths", 1)
        device = lengths.device
    out_len = torch.empty((batch,), dtype=torch.int64, device=device)
    mask = torch.zeros((batch, t_valid), dtype=torch.uint8, device=device)
    klen = torch.empty((batch,), dtype=torch.int32, device=device)
    N.check(N.lib().fcenc_lengths(N.ptr(lengths), N.ptr(out_len), N.ptr(mask), N.ptr(klen), batch, t_valid,
                                  torch.cuda.current_stream(device).cuda_stream))
    return out_len, mask.view(torch.bool), klen


@output_lengths.register_fake
def _(lengths, batch, t_valid, device):
    dev = lengths.device if lengths is not None else device
    return (torch.empty((batch,), dtype=torch.int64, device=dev),
            torch.empty((batch, t_valid), dtype=torch.bool, device=dev),
            torch.empty((batch,), dtype=torch.int32, device=dev))


@torch.library.custom_op("fcenc::ffn_pack", mutates_args=())
def ffn_pack(w1: Tensor, w2: Tensor) -> Tuple[Tensor, Tensor]:
    """Row-major bf16 Linear weights w1 [ff][d], w2 [d][ff] -> the fragment order ffn_fused_ streams
    (same shapes, different element order)."""
    N.require(w1, torch.bfloat16, "w1", 2)
    N.require(w2, torch.bfloat16, "w2", 2)
    ff, d = w1.shape
    if tuple(w2.shape) != (d, ff):
        raise ValueError("fcenc::ffn_pack: weight shapes must be [ff][d] and [d][ff]")
    w1p, w2p = torch.empty_like(w1), torch.empty_like(w2)
    N.check(N.lib().fcenc_ffn_pack(N.ptr(w1), N.ptr(w2), N.ptr(w1p), N.ptr(w2p), d, ff, N.stream_of(w1)))
    return w1p, w2p


@ffn_pack.register_fake
def _(w1, w2):
    return torch.empty_like(w1), torch.empty_like(w2)


@torch.library.custom_op("fcenc::ffn_fused_", mutates_args=("x", "xc"))
def ffn_fused_(x: Tensor, ln_g: Tensor, ln_b: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor,
               post_g: Optional[Tensor], post_b: Optional[Tensor], alpha: float, xc: Optional[Tensor]) -> None:
    """x += alpha * (silu(LN(x) @ w1.T + b1) @ w2.T + b2), then optionally x = LN(x; post_g, post_b);
    bf16 weights as returned by ffn_pack, d = 256 only (one kernel, the hidden activation stays on chip)."""
    N.require(x, torch.float32, "x", 2)
    N.require(w1, torch.bfloat16, "w1", 2)
    N.require(w2, torch.bfloat16, "w2", 2)
    for nm, t in (("ln_g", ln_g), ("ln_b", ln_b), ("b1", b1), ("b2", b2), ("post_g", post_g), ("post_b", post_b)):
        if t is not None:
            N.require(t, torch.float32, nm, 1)
    m, d = x.shape
    ff = w1.shape[0]
    if tuple(w1.shape) != (ff, d) or tuple(w2.shape) != (d, ff):
        raise ValueError("fcenc::ffn_fused_: weight shapes must be [ff][d] and [d][ff]")
    if xc is not None:
        N.require(xc, torch.bfloat16, "xc", 2)
    N.check(N.lib().fcenc_ffn_fused(N.ptr(x), N.ptr(ln_g), N.ptr(ln_b), N.ptr(w1), N.ptr(b1), N.ptr(w2), N.ptr(b2),
                                    N.ptr(post_g), N.ptr(post_b), N.ptr(xc), None, None, None, None, 0, m, d, ff,
                                    0, 0, alpha, N.stream_of(x)))


@torch.library.custom_op("fcenc::rows_pack", mutates_args=())
def rows_pack(w: Tensor) -> Tensor:
    """Row-major bf16 [rows][256] (rows % 256 == 0) -> MFMA fragment order by 256-row block."""
    N.require(w, torch.bfloat16, "w", 2)
    wp = torch.empty_like(w)
    N.check(N.lib().fcenc_rows_pack(N.ptr(w), N.ptr(wp), w.shape[0], w.shape[1], N.stream_of(w)))
    return wp


@rows_pack.register_fake
def _(w):
    return torch.empty_like(w)


@torch.library.custom_op("fcenc::qkv_pack", mutates_args=())
def qkv_pack(w: Tensor) -> Tensor:
    """Row-major bf16 q | k | v projection matrix [768][256] -> the paired fragment order ffn_qkv_fused_ and the
    block kernels stream (a lane's two fragments = 8 consecutive output features)."""
    N.require(w, torch.bfloat16, "w", 2)
    wp = torch.empty_like(w)
    N.check(N.lib().fcenc_qkv_pack(N.ptr(w), N.ptr(wp), w.shape[0], w.shape[1], N.stream_of(w)))
    return wp


@qkv_pack.register_fake
def _(w):
    return torch.empty_like(w)


@torch.library.custom_op("fcenc::ffn_qkv_fused_", mutates_args=("x",))
def ffn_qkv_fused_(x: Tensor, ln_g: Tensor, ln_b: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor,
                   alpha: float, wq: Tensor, bq: Tensor) -> Tensor:
    """ffn_fused_ followed, in the same kernel, by the projection of the new x: returns bf16
    x @ wq.T + bq with wq [768][256] as returned by qkv_pack."""
    N.require(x, torch.float32, "x", 2)
    for nm, t in (("w1", w1), ("w2", w2), ("wq", wq)):
        N.require(t, torch.bfloat16, nm, 2)
    for nm, t in (("ln_g", ln_g), ("ln_b", ln_b), ("b1", b1), ("b2", b2), ("bq", bq)):
        N.require(t, torch.float32, nm, 1)
    m, d = x.shape
    ff, nq = w1.shape[0], wq.shape[0]
    q = torch.empty((m, nq), dtype=torch.bfloat16, device=x.device)
    N.check(N.lib().fcenc_ffn_fused(N.ptr(x), N.ptr(ln_g), N.ptr(ln_b), N.ptr(w1), N.ptr(b1), N.ptr(w2), N.ptr(b2),
                                    None, None, None, None, N.ptr(wq), N.ptr(bq), N.ptr(q), nq, m, d, ff, 0, 0,
                                    alpha, N.stream_of(x)))
    return q


@ffn_qkv_fused_.register_fake
def _(x, ln_g, ln_b, w1, b1, w2, b2, alpha, wq, bq):
    return x.new_empty((x.shape[0], wq.shape[0]), dtype=torch.bfloat16)


@torch.library.custom_op("fcenc::convmod_pack", mutates_args=())
def convmod_pack(w1: Tensor, w2: Tensor) -> Tuple[Tensor, Tensor]:
    """Row-major bf16 pointwise weights w1 [2d][d] (GLU-interleaved rows), w2 [d][d] -> the order
    convmod_fused streams (same shapes, different element order)."""
    N.require(w1, torch.bfloat16, "w1", 2)
    N.require(w2, torch.bfloat16, "w2", 2)
    d = w2.shape[0]
    if tuple(w1.shape) != (2 * d, d) or tuple(w2.shape) != (d, d):
        raise ValueError("fcenc::convmod_pack: weight shapes must be [2d][d] and [d][d]")
    w1p, w2p = torch.empty_like(w1), torch.empty_like(w2)
    N.check(N.lib().fcenc_convmod_pack(N.ptr(w1), N.ptr(w2), N.ptr(w1p), N.ptr(w2p), d, N.stream_of(w1)))
    return w1p, w2p


@convmod_pack.register_fake
def _(w1, w2):
    return torch.empty_like(w1), torch.empty_like(w2)


@torch.library.custom_op("fcenc::convmod_fused", mutates_args=())
def convmod_fused(x: Tensor, ln_g: Tensor, ln_b: Tensor, w1: Tensor, b1: Tensor, dw_w9: Tensor, dw_b: Tensor,
                  w2: Tensor, b2: Tensor, padded_frames: int, attn: Optional[Tensor] = None,
                  wo: Optional[Tensor] = None, bo: Optional[Tensor] = None) -> Tensor:
    """x + pw2(silu(dw9(glu(pw1(LN(x)))))) for x fp32 [B * padded_frames][256] (one kernel; bf16 weights,
    w1 / w2 as returned by convmod_pack, b1 in the GLU-interleaved row order, BatchNorm folded into
    dw_w9 / dw_b).  With attn (bf16 [M][256]), wo (rows_pack'ed) and bo the kernel first replaces x by
    x + attn @ wo.T + bo.  Returns a new tensor."""
    if attn is not None:
        N.require(attn, torch.bfloat16, "attn", 2)
        N.require(wo, torch.bfloat16, "wo", 2)
        N.require(bo, torch.float32, "bo", 1)
    N.require(x, torch.float32, "x", 2)
    N.require(w1, torch.bfloat16, "w1", 2)
    N.require(w2, torch.bfloat16, "w2", 2)
    N.require(dw_w9, torch.float32, "dw_w9", 2)
    for nm, t in (("ln_g", ln_g), ("ln_b", ln_b), ("b1", b1), ("dw_b", dw_b), ("b2", b2)):
        N.require(t, torch.float32, nm, 1)
    m, d = x.shape
    if tuple(w1.shape) != (2 * d, d) or tuple(w2.shape) != (d, d) or tuple(dw_w9.shape) != (9, d):
        raise ValueError("fcenc::convmod_fused: weight shapes must be [2d][d], [9][d] and [d][d]")
    out = torch.empty_like(x)
    N.check(N.lib().fcenc_convmod_fused(N.ptr(x), N.ptr(out), N.ptr(ln_g), N.ptr(ln_b), N.ptr(w1), N.ptr(b1),
                                        N.ptr(dw_w9), N.ptr(dw_b), N.ptr(w2), N.ptr(b2), N.ptr(attn), N.ptr(wo),
                                        N.ptr(bo), m, d, padded_frames, N.stream_of(x)))
    return out


@convmod_fused.register_fake
def _(x, ln_g, ln_b, w1, b1, dw_w9, dw_b, w2, b2, padded_frames, attn=None, wo=None, bo=None):
    return torch.empty_like(x)


@torch.library.custom_op("fcenc::stream_attention_", mutates_args=("kcache", "vcache"))
def stream_attention_(qkv: Tensor, pos: Tensor, bias_u: Tensor, bias_v: Tensor, kcache: Tensor, vcache: Tensor,
                      frames: int, scale: float) -> Tensor:
    """Attention of a chunk over the cached keys (model.py:352-428 with cache).  qkv [B*Tq][3*H*64] (tokens
    packed stream-major), pos [2*Tq-1][H*64], rings kcache [B][H][cap][64] and vcache [B][H][cap/8][64][8]
    (frame t in slot t % cap; V in blocks of 8 slots with one 8-slot group per dimension) that receive the
    chunk's rows; `frames` = frames every stream has seen before -> [B*Tq][H*64]."""
    N.require(qkv, qkv.dtype, "qkv", 2)
    N.require(pos, qkv.dtype, "pos", 2)
    N.require(bias_u, torch.float32, "bias_u", 1)
    N.require(bias_v, torch.float32, "bias_v", 1)
    N.require(kcache, qkv.dtype, "kcache", 4)
    N.require(vcache, qkv.dtype, "vcache", 5)
    batch, heads, cap, _ = kcache.shape
    hd = heads * 64
    tq = qkv.shape[0] // batch
    if (qkv.shape[0] != batch * tq or qkv.shape[1] != 3 * hd or kcache.shape[3] != 64 or cap % 8
            or tuple(vcache.shape) != (batch, heads, cap // 8, 64, 8)):
        raise ValueError("fcenc::stream_attention_: qkv must be [B*Tq][3*H*64], kcache [B][H][cap][64], "
                         "vcache [B][H][cap/8][64][8]")
    if tuple(pos.shape) != (2 * tq - 1, hd) or bias_u.numel() != hd or bias_v.numel() != hd:
        raise ValueError("fcenc::stream_attention_: operand shapes do not match qkv")
    out = torch.empty((batch * tq, hd), dtype=qkv.dtype, device=qkv.device)
    N.check(N.lib().fcenc_stream_attention(_cdt(qkv), N.ptr(qkv), N.ptr(pos), N.ptr(bias_u), N.ptr(bias_v),
                                           N.ptr(kcache), N.ptr(vcache), N.ptr(out), batch, hd // 64, tq, cap, frames,
                                           scale, N.stream_of(qkv)))
    return out


@stream_attention_.register_fake
def _(qkv, pos, bias_u, bias_v, kcache, vcache, frames, scale):
    return qkv.new_empty((qkv.shape[0], qkv.shape[1] // 3))


@torch.library.custom_op("fcenc::stream_dwconv9_bn_silu_", mutates_args=("cache",))
def stream_dwconv9_bn_silu_(g: Tensor, cache: Tensor, w9: Tensor, bias: Tensor) -> Tensor:
    """g [B][Tq][d] chunk, cache [B][8][d] frames before it (replaced by the last 8 frames of cache | chunk):
    causal depthwise conv k=9 + folded BatchNorm + SiLU (model.py:236-249, :523-538)."""
    N.require(g, g.dtype, "g", 3)
    N.require(cache, g.dtype, "cache", 3)
    N.require(w9, torch.float32, "w9", 2)
    N.require(bias, torch.float32, "bias", 1)
    b, tq, d = g.shape
    if tuple(cache.shape) != (b, 8, d) or tuple(w9.shape) != (9, d) or bias.numel() != d:
        raise ValueError("fcenc::stream_dwconv9_bn_silu_: cache must be [B][8][d], w9 [9][d], bias [d]")
    out = torch.empty_like(g)
    N.check(N.lib().fcenc_stream_dwconv(_cdt(g), N.ptr(g), N.ptr(cache), N.ptr(w9), N.ptr(bias), N.ptr(out), b, tq, d,
                                        N.stream_of(g)))
    return out


@stream_dwconv9_bn_silu_.register_fake
def _(g, cache, w9, bias):
    return torch.empty_like(g)


@torch.library.custom_op("fcenc::greedy_decode", mutates_args=())
def greedy_decode(enc_proj: Tensor, pred_proj: Tensor, out_w: Tensor, out_b: Tensor, d_joint: int, blank: int,
                  max_symbols: int, clamp: float, capacity: int, last: Optional[Tensor] = None
                  ) -> Tuple[Tensor, Tensor, Tensor]:
    """enc_proj [B][T][ld] fp32 (first d_joint columns used), pred_proj / out_w [V][d_joint], out_b [V],
    last [B] int32 or None (start from token 0) -> (tokens int32 [B][capacity], counts int32 [B], last int32 [B]
    = the token each utterance ended on): the greedy loop of train.py:180-211 on the device, resumable across
    chunks like inference.py:379-406."""
    N.require(enc_proj, torch.float32, "enc_proj", 3)
    N.require(pred_proj, torch.float32, "pred_proj", 2)
    N.require(out_w, torch.float32, "out_w", 2)
    N.require(out_b, torch.float32, "out_b", 1)
    b, t, ld = enc_proj.shape
    v = out_w.shape[0]
    if tuple(out_w.shape) != (v, d_joint) or tuple(pred_proj.shape) != (v, d_joint) or out_b.numel() != v:
        raise ValueError("fcenc::greedy_decode: pred_proj / out_w must be [V][d_joint], out_b [V]")
    if last is None:
        last_out = torch.zeros((b,), dtype=torch.int32, device=enc_proj.device)
    else:
        N.require(last, torch.int32, "last", 1)
        if last.numel() != b:
            raise ValueError("fcenc::greedy_decode: last must hold one token per utterance")
        last_out = last.clone()
    tokens = torch.empty((b, capacity), dtype=torch.int32, device=enc_proj.device)
    counts = torch.empty((b,), dtype=torch.int32, device=enc_proj.device)
    N.check(N.lib().fcenc_greedy_decode(N.ptr(enc_proj), ld, N.ptr(pred_proj), N.ptr(out_w), N.ptr(out_b), b, t,
                                        d_joint, v, blank, max_symbols, clamp, N.ptr(tokens), capacity, N.ptr(counts),
                                        N.ptr(last_out), N.stream_of(enc_proj)))
    return tokens, counts, last_out


@greedy_decode.register_fake
def _(enc_proj, pred_proj, out_w, out_b, d_joint, blank, max_symbols, clamp, capacity, last=None):
    b = enc_proj.shape[0]
    return (torch.empty((b, capacity), dtype=torch.int32, device=enc_proj.device),
            torch.empty((b,), dtype=torch.int32, device=enc_proj.device),
            torch.empty((b,), dtype=torch.int32, device=enc_proj.device))
