"""ctypes binding of libfcenc_hip.so (C ABI in include/fcenc.h).

The library is built in-tree by `python -m conformer_amd.build`; nothing is compiled at
import.  Loading fails loudly: there is no eager / CPU substitute for these kernels.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FCENC_LIB_PATH") or os.path.join(_HERE, "libfcenc_hip.so")   # override: A/B builds

F32, BF16 = 0, 1
EPI_STORE, EPI_STORE_SILU, EPI_GLU, EPI_RESID = 0, 1, 2, 3

_vp, _fp, _i, _f = C.c_void_p, C.c_void_p, C.c_int, C.c_float


class Layer(C.Structure):
    _names = ["ff1_ln_g", "ff1_ln_b", "ff1_w1", "ff1_b1", "ff1_w2", "ff1_b2", "qkv_w", "qkv_b", "pos",
              "bias_u", "bias_v", "out_w", "out_b", "conv_ln_g", "conv_ln_b", "pw1_w", "pw1_b", "dw_w9", "dw_b",
              "pw2_w", "pw2_b", "ff2_ln_g", "ff2_ln_b", "ff2_w1", "ff2_b1", "ff2_w2", "ff2_b2", "ln_g", "ln_b"]
    _fields_ = [(n, C.c_void_p) for n in _names]


class Encoder(C.Structure):
    _fields_ = [("dtype", C.c_int), ("d_model", C.c_int), ("n_heads", C.c_int), ("head_dim", C.c_int),
                ("n_layers", C.c_int), ("ff_dim", C.c_int), ("channels", C.c_int), ("n_mels", C.c_int),
                ("window", C.c_int),
                ("conv1_w9", C.c_void_p), ("conv1_b", C.c_void_p), ("conv2_w", C.c_void_p), ("conv2_b", C.c_void_p),
                ("conv3_w", C.c_void_p), ("conv3_b", C.c_void_p), ("lin_w", C.c_void_p), ("lin_b", C.c_void_p),
                ("layers", C.POINTER(Layer)), ("ev_begin", C.c_void_p), ("ev_end", C.c_void_p),
                ("ev_block_begin", C.c_void_p), ("ev_block_end", C.c_void_p),
                ("conv12_a1", C.c_void_p), ("conv12_w2", C.c_void_p), ("conv3_wp", C.c_void_p)]


class StreamLayer(C.Structure):
    _names = ["kcache", "vcache", "conv_cache", "pos", "out_w", "pw1_w", "pw2_w", "qkv_w"]
    _fields_ = [(n, C.c_void_p) for n in _names]


class StreamState(C.Structure):
    _fields_ = [("batch", C.c_int), ("cap", C.c_int), ("frames", C.c_longlong), ("layers", C.POINTER(StreamLayer)),
                ("ring", C.c_void_p)]


class MelPlan(C.Structure):
    _fields_ = [("n_fft", C.c_int), ("hop", C.c_int), ("kp", C.c_int), ("wpitch", C.c_int), ("n_mels", C.c_int),
                ("fb_maxw", C.c_int), ("dft", C.c_void_p), ("win", C.c_void_p), ("fb_start", C.c_void_p),
                ("fb_count", C.c_void_p), ("fb_w", C.c_void_p)]


class StreamLanes(C.Structure):
    _fields_ = [("n", C.c_int), ("streams", C.c_void_p * 7), ("events", C.c_void_p * 8)]


_SIGNATURES = {
    "fcenc_version": (C.c_int, []),
    "fcenc_last_error": (C.c_char_p, []),
    "fcenc_gemm": (C.c_int, [_i, _i, _vp, _vp, _fp, _vp, _fp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp]),
    "fcenc_conv3x3s2": (C.c_int, [_i, _vp, _vp, _fp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fcenc_input_stats": (C.c_int, [_fp, _vp, _i, _i, _i, _i, _vp]),
    "fcenc_conv1": (C.c_int, [_i, _fp, _vp, _i, _fp, _fp, _vp, _i, _i, _i, _i, _vp]),
    "fcenc_conv12_supported": (C.c_int, [_i, _i, _i]),
    "fcenc_conv12_packed_bytes": (C.c_longlong, [_i]),
    "fcenc_conv12_pack": (C.c_int, [_fp, _fp, _vp, _vp, _vp, _i, _i, _vp]),
    "fcenc_lognorm_conv12": (C.c_int, [_fp, _vp, _i, _vp, _vp, _fp, _vp, _i, _i, _i, _i, _vp]),
    "fcenc_conv3_patch_supported": (C.c_int, [_i, _i, _i, _i]),
    "fcenc_conv3_patch_packed_bytes": (C.c_longlong, []),
    "fcenc_conv3_patch_pack": (C.c_int, [_vp, _vp, _i, _vp]),
    "fcenc_conv3_patch": (C.c_int, [_vp, _vp, _fp, _vp, _i, _i, _i, _i, _i, _vp]),
    "fcenc_lognorm_conv1_static": (C.c_int, [_i, _fp, _fp, _fp, _vp, _i, _i, _i, _i, _vp]),
    "fcenc_layernorm": (C.c_int, [_i, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _fp, _i, _i, _i, _i, _vp]),
    "fcenc_layernorm_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "fcenc_glu_dwconv_fwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "fcenc_glu_dwconv_partial_blocks": (C.c_int, [_i, _i]),
    "fcenc_glu_dwconv_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "fcenc_linear_bwd_workspace_bytes": (C.c_longlong, [_i, _i, _i, _i]),
    "fcenc_linear_bwd": (C.c_int, [_i, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _i, _i, _i, _vp]),
    "fcenc_bn_partial_blocks": (C.c_int, [_i]),
    "fcenc_bn_silu_train_fwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _f, _f, _fp, _fp, _fp, _i, _i, _vp]),
    "fcenc_bn_silu_train_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _vp]),
    "fcenc_ffn_fused_supported": (C.c_int, [_i, _i, _i]),
    "fcenc_ffn_pack": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "fcenc_ffn_qkv_fused_supported": (C.c_int, [_i, _i, _i, _i]),
    "fcenc_rows_pack": (C.c_int, [_vp, _vp, _i, _i, _vp]),
    "fcenc_qkv_pack": (C.c_int, [_vp, _vp, _i, _i, _vp]),
    "fcenc_ffn_fused": (C.c_int, [_fp, _fp, _fp, _vp, _fp, _vp, _fp, _fp, _fp, _vp, _fp, _vp, _fp, _vp, _i,
                                  _i, _i, _i, _i, _i, _f, _vp]),
    "fcenc_convmod_fused_supported": (C.c_int, [_i, _i, _i]),
    "fcenc_convmod_pack": (C.c_int, [_vp, _vp, _vp, _vp, _i, _vp]),
    "fcenc_convmod_fused": (C.c_int, [_fp, _fp, _fp, _fp, _vp, _fp, _fp, _fp, _vp, _fp, _vp, _vp, _fp, _i, _i, _i,
                                      _vp]),
    "fcenc_convmod_oproj_supported": (C.c_int, [_i, _i, _i, _i]),
    "fcenc_block_tail": (C.c_int, [C.POINTER(Layer), _fp, _fp, _vp, _fp, _i, _i, _i, _i, _i, _i, _vp]),
    "fcenc_dwconv": (C.c_int, [_i, _vp, _fp, _fp, _vp, _i, _i, _i, _vp]),
    "fcenc_attention": (C.c_int, [_i, _vp, _vp, _fp, _fp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "fcenc_lengths": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "fcenc_encoder_workspace_bytes": (C.c_size_t, [C.POINTER(Encoder), _i, _i]),
    "fcenc_encoder_forward": (C.c_int, [C.POINTER(Encoder), _fp, _vp, _i, _i, _fp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "fcenc_encoder_forward_lanes": (C.c_int, [C.POINTER(Encoder), _fp, _vp, _i, _i, _fp, _vp, _vp, _vp, C.c_size_t, _vp,
                                              C.POINTER(StreamLanes)]),
    "fcenc_stream_attention": (C.c_int, [_i, _vp, _vp, _fp, _fp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_longlong, _f, _vp]),
    "fcenc_stream_dwconv": (C.c_int, [_i, _vp, _vp, _fp, _fp, _vp, _i, _i, _i, _vp]),
    "fcenc_stream_max_chunk_frames": (C.c_int, []),
    "fcenc_stream_max_cache_frames": (C.c_int, []),
    "fcenc_encoder_stream_workspace_bytes": (C.c_size_t, [C.POINTER(Encoder), _i, _i, _i]),
    "fcenc_encoder_stream_step": (C.c_int, [C.POINTER(Encoder), C.POINTER(StreamState), _fp, _i, _fp, _vp, C.c_size_t,
                                            _vp, C.POINTER(StreamLanes)]),
    "fcenc_greedy_decode": (C.c_int, [_fp, _i, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp, _vp]),
    "fcenc_mel_spectrogram": (C.c_int, [C.POINTER(MelPlan), _fp, C.c_longlong, _vp, _i, _i, _fp, _i, _vp, _vp]),
    "fcenc_out_frames": (C.c_int, [_i]),
    "fcenc_padded_frames": (C.c_int, [_i]),
    "fcenc_event_create": (C.c_void_p, []),
    "fcenc_event_destroy": (None, [_vp]),
    "fcenc_event_record": (C.c_int, [_vp, _vp]),
    "fcenc_event_elapsed_ms": (C.c_float, [_vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m conformer_amd.build` "
                "(hipcc, gfx950). The encoder has no non-HIP execution path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError if the ABI and the header diverge
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("fcenc: " + lib().fcenc_last_error().decode())


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise ValueError(f"fcenc kernels compute in float32 or bfloat16, not {dt}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_of(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def require(t: torch.Tensor, dtype: torch.dtype, name: str, ndim: Optional[int] = None) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"fcenc: {name} must live on the GPU (got {t.device})")
    if t.dtype != dtype:
        raise TypeError(f"fcenc: {name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"fcenc: {name} must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"fcenc: {name} must have {ndim} dims, got {tuple(t.shape)}")
    if t.data_ptr() % 16:
        raise ValueError(f"fcenc: {name} must be 16-byte aligned")
    return t
