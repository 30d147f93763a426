"""Drop-in module tree of the FastConformer-Transducer for MI355X.

Same constructor arguments, attribute names, parameter/buffer names and registration order
as the reference (`/root/reference/model.py:1015-1092`), so `create_model(config)`,
`load_state_dict(checkpoint['model_state_dict'])` and a given `torch.manual_seed` produce the
same model.  What differs is how the encoder runs:

  * eval mode on a GPU tensor (validation `train.py:233`, inference `inference.py:139`):
    `FastConformerEncoder.forward` hands the whole pass to libfcenc_hip.so (one C call, all
    HIP kernels) -- there is no eager substitute; a missing library raises;
  * eval mode with `streaming=True` (`inference.py:358`): one C call per chunk; the per-layer
    caches live in device ring buffers held by a `StreamCache` (conformer_amd/streaming.py) that is
    returned as `new_cache` and updated IN PLACE by the next chunk (the reference returns fresh
    tensors; keep `cache.to_reference()` if an older state is needed again);
  * training mode needs autograd: the module tree below runs op by op, with the matrix products,
    LayerNorms and the GLU + depthwise core on the HIP kernels of conformer_amd.autograd when the
    tensors are on the GPU (`encoder.native_training`, default on) and on ATen otherwise; a streaming
    call the kernels are not built for (lengths given, chunks of more than min(16, local_context) output frames) takes
    the same op-by-op path in eval mode, on ATen, with a warning.

The predictor and joint networks are plain PyTorch (outside the accelerated path).
"""
from __future__ import annotations

import math
import threading
import warnings
import random
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .streaming import StreamCache

Tensor = torch.Tensor


# ---- training on the HIP kernels -------------------------------------------------------------------------
# While a training-mode forward of an encoder with `native_training` runs on the GPU, the Linear / 1x1-conv products,
# the LayerNorms and the GLU + depthwise core of the conv module below go through conformer_amd.autograd (HIP
# kernels with backward) instead of ATen.  The switch is per thread and only ever set by forward_composite.
_native = threading.local()


def _native_on() -> bool:
    return getattr(_native, "on", False)


def _lin(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """F.linear, or the HIP GEMM with autograd on the training path (weight (N, K) or a 1x1 Conv1d's (N, K, 1))."""
    w2 = weight.reshape(weight.shape[0], -1)
    if _native_on() and x.is_cuda and w2.shape[0] % 128 == 0 and w2.shape[1] % 128 == 0:
        from . import autograd as A
        return A.linear(x, w2, bias, getattr(_native, "bf16", False))
    return F.linear(x, w2, bias)


def _ln(x: Tensor, ln: nn.LayerNorm) -> Tensor:
    if _native_on() and x.is_cuda and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024:
        from . import autograd as A
        return A.layer_norm(x, ln.weight, ln.bias)
    return ln(x)


def _is_fake(t: Tensor) -> bool:
    from torch._subclasses.fake_tensor import is_fake
    return is_fake(t)


Cache = List[Tuple[Optional[Tensor], Optional[Tensor]]]


def stochastic_depth_schedule(num_layers: int, drop_prob: float, mode: str = "linear", start_layer: int = 1
                              ) -> List[float]:
    """Per-layer skip probabilities: zero before `start_layer` (1-indexed), then either a linear
    ramp up to `drop_prob` or a constant (model.py:59-93)."""
    if not 0 <= drop_prob < 1.0:
        raise ValueError("drop_prob must be in [0, 1)")
    if mode not in ("linear", "uniform"):
        raise ValueError(f"Unknown mode: {mode}")
    n_active = max(num_layers - start_layer + 1, 0)
    ramp = [drop_prob * (k + 1) / n_active if mode == "linear" else drop_prob for k in range(n_active)]
    return [0.0] * (start_layer - 1) + ramp


class RelativePositionalEncoding(nn.Module):
    """Holds the sinusoid table `pe` (persistent buffer, (2*max_len-1, d), row p+max_len-1 for
    relative position p) and scales the input by sqrt(d)   (model.py:109-182)."""

    def __init__(self, d_model: int, max_len: int = 5000, dropout: float = 0.1, scale: bool = True):
        super().__init__()
        self.d_model, self.max_len = d_model, max_len
        self.dropout = nn.Dropout(dropout)
        self.scale = math.sqrt(d_model) if scale else 1.0
        self.register_buffer("pe", self._sinusoids(max_len))

    def _sinusoids(self, max_len: int) -> Tensor:
        pos = torch.arange(-(max_len - 1), max_len, dtype=torch.float32).unsqueeze(1)
        freq = torch.exp(torch.arange(0, self.d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / self.d_model))
        pe = torch.zeros(2 * max_len - 1, self.d_model)
        pe[:, 0::2] = torch.sin(pos * freq)
        pe[:, 1::2] = torch.cos(pos * freq)
        return pe

    def table(self, seq_len: int) -> Tensor:
        """Rows for relative positions -(seq_len-1)..(seq_len-1); grows the buffer when needed."""
        if seq_len > self.max_len:
            self.max_len = max(seq_len, 2 * self.max_len)
            self.pe = self._sinusoids(self.max_len).to(self.pe.device)
        mid = self.max_len - 1
        return self.pe[mid - seq_len + 1: mid + seq_len]

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        return self.dropout(x * self.scale), self.table(x.size(1))


class CausalConv1d(nn.Module):
    """Conv1d that only looks back: left context comes from `cache` (previous chunk) or zeros;
    returns the last (k-1)*dilation input frames as the next cache (model.py:189-256)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1,
                 dilation: int = 1, groups: int = 1, bias: bool = True):
        super().__init__()
        self.kernel_size, self.dilation, self.stride = kernel_size, dilation, stride
        self.cache_size = (kernel_size - 1) * dilation
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=0,
                              dilation=dilation, groups=groups, bias=bias)

    def forward(self, x: Tensor, cache: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        x = F.pad(x, (self.cache_size, 0)) if cache is None else torch.cat([cache, x], dim=-1)
        tail = x[..., x.size(-1) - self.cache_size:].clone() if self.cache_size > 0 else None
        return self.conv(x), tail

    def get_initial_cache(self, batch_size: int, device: torch.device) -> Tensor:
        return torch.zeros(batch_size, self.conv.in_channels, self.cache_size, device=device)


class RelPositionMultiHeadAttention(nn.Module):
    """Transformer-XL attention: content term (q+u).k plus position term (q+v).p with the
    relative shift, optional causal sliding window and key-padding mask (model.py:263-431)."""

    def __init__(self, d_model: int, num_heads: int, dropout: float = 0.1, max_pos_encoding: int = 5000,
                 use_local_attn: bool = False, local_context: int = 256):
        super().__init__()
        assert d_model % num_heads == 0
        self.d_model, self.num_heads = d_model, num_heads
        self.d_head = d_model // num_heads
        self.scale = self.d_head ** -0.5
        self.use_local_attn, self.local_context = use_local_attn, local_context
        self.q_proj = nn.Linear(d_model, d_model)
        self.k_proj = nn.Linear(d_model, d_model)
        self.v_proj = nn.Linear(d_model, d_model)
        self.out_proj = nn.Linear(d_model, d_model)
        self.pos_proj = nn.Linear(d_model, d_model, bias=False)
        self.pos_bias_u = nn.Parameter(torch.zeros(num_heads, self.d_head))
        self.pos_bias_v = nn.Parameter(torch.zeros(num_heads, self.d_head))
        nn.init.xavier_uniform_(self.pos_bias_u.unsqueeze(0))
        nn.init.xavier_uniform_(self.pos_bias_v.unsqueeze(0))
        self.dropout = nn.Dropout(dropout)

    @staticmethod
    def _rel_shift(x: Tensor) -> Tensor:
        # (b,h,q,w): prepend a zero column, reinterpret the flat data as (w+1, q), drop the first row
        b, h, q, w = x.shape
        return F.pad(x, (1, 0)).view(b, h, w + 1, q)[:, :, 1:].reshape(b, h, q, w)

    def forward(self, x: Tensor, pos_emb: Tensor, mask: Optional[Tensor] = None, cache: Optional[Tensor] = None
                ) -> Tuple[Tensor, Optional[Tensor]]:
        bsz, tq, _ = x.shape
        h, dh = self.num_heads, self.d_head
        q, k, v = (_lin(x, p.weight, p.bias) for p in (self.q_proj, self.k_proj, self.v_proj))
        if cache is not None:
            past_k, past_v = cache.chunk(2, dim=-1)
            k, v = torch.cat([past_k, k], 1), torch.cat([past_v, v], 1)
            if k.size(1) > self.local_context:
                k, v = k[:, -self.local_context:], v[:, -self.local_context:]
        kv_cache = torch.cat([k, v], dim=-1)
        kv = k.size(1)
        q = q.view(bsz, tq, h, dh)
        k = k.view(bsz, kv, h, dh).permute(0, 2, 3, 1)
        v = v.view(bsz, kv, h, dh).transpose(1, 2)
        p = _lin(pos_emb, self.pos_proj.weight, None).view(-1, h, dh).permute(1, 2, 0)
        content = torch.matmul((q + self.pos_bias_u).transpose(1, 2), k)
        position = self._rel_shift(torch.matmul((q + self.pos_bias_v).transpose(1, 2), p))
        if position.size(-1) > kv:
            position = position[..., :kv]
        elif position.size(-1) < kv:
            position = F.pad(position, (kv - position.size(-1), 0))
        score = (content + position) * self.scale
        if self.use_local_attn and kv > self.local_context:
            qi = torch.arange(tq, device=x.device).unsqueeze(1) + (kv - tq)
            kj = torch.arange(kv, device=x.device).unsqueeze(0)
            score = score.masked_fill((kj > qi) | (kj <= qi - self.local_context), float("-inf"))
        if mask is not None:
            score = score.masked_fill(mask[:, None, None, :] if mask.dim() == 2 else mask, float("-inf"))
        w = self.dropout(F.softmax(score, dim=-1))
        ctx = torch.matmul(w, v).transpose(1, 2).reshape(bsz, tq, self.d_model)
        return _lin(ctx, self.out_proj.weight, self.out_proj.bias), kv_cache


class ConformerConvolution(nn.Module):
    """LayerNorm -> pointwise (d->2d) -> GLU -> depthwise (causal) -> norm -> SiLU -> pointwise
    (model.py:438-550)."""

    def __init__(self, d_model: int, kernel_size: int = 9, dropout: float = 0.1, norm_type: str = "batch",
                 use_causal: bool = True):
        super().__init__()
        self.use_causal, self.kernel_size = use_causal, kernel_size
        self.layer_norm = nn.LayerNorm(d_model)
        self.pointwise1 = nn.Conv1d(d_model, 2 * d_model, 1)
        self.glu = nn.GLU(dim=1)
        if use_causal:
            self.depthwise = CausalConv1d(d_model, d_model, kernel_size, groups=d_model)
        else:
            self.depthwise = nn.Conv1d(d_model, d_model, kernel_size, padding=(kernel_size - 1) // 2, groups=d_model)
        norms = {"batch": lambda: nn.BatchNorm1d(d_model), "layer": lambda: nn.LayerNorm(d_model),
                 "instance": lambda: nn.InstanceNorm1d(d_model), "group": lambda: nn.GroupNorm(8, d_model)}
        self.norm = norms.get(norm_type, nn.Identity)()
        self.norm_type = norm_type
        self.activation = nn.SiLU()
        self.pointwise2 = nn.Conv1d(d_model, d_model, 1)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x: Tensor, cache: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        if (_native_on() and x.is_cuda and cache is None and not getattr(_native, "streaming", False) and self.use_causal
                and self.kernel_size == 9 and self.norm_type == "batch" and x.shape[-1] % 128 == 0):
            # token-major throughout: no transposes; GLU + causal depthwise conv in one HIP pass each way, BatchNorm
            # over (batch x time) per channel with batch statistics + SiLU on the HIP kernels, pointwise convs as GEMMs
            from . import autograd as A
            y = _lin(_ln(x, self.layer_norm), self.pointwise1.weight, self.pointwise1.bias)            # (B, T, 2d)
            z = A.glu_dwconv(y, self.depthwise.conv.weight, self.depthwise.conv.bias)                  # (B, T, d)
            if A.bn_silu_ok(self.norm, z.shape[-1]):
                z = A.bn_silu(z, self.norm)                         # batch statistics + SiLU, running stats updated
            else:
                z = self.activation(self.norm(z.reshape(-1, z.shape[-1])).view(z.shape))
            z = _lin(z, self.pointwise2.weight, self.pointwise2.bias)
            return self.dropout(z), None
        y = self.glu(self.pointwise1(self.layer_norm(x).transpose(1, 2)))
        new_cache = None
        if self.use_causal:
            y, new_cache = self.depthwise(y, cache)
        else:
            y = self.depthwise(y)
        y = self.norm(y.transpose(1, 2)).transpose(1, 2) if self.norm_type == "layer" else self.norm(y)
        y = self.dropout(self.pointwise2(self.activation(y)))
        return y.transpose(1, 2), new_cache

    def get_initial_cache(self, batch_size: int, device: torch.device) -> Optional[Tensor]:
        return self.depthwise.get_initial_cache(batch_size, device) if self.use_causal else None


class FeedForwardModule(nn.Module):
    """LayerNorm -> Linear(d, e*d) -> SiLU -> Linear(e*d, d)   (model.py:557-584)."""

    def __init__(self, d_model: int, expansion_factor: int = 4, dropout: float = 0.1):
        super().__init__()
        self.layer_norm = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_model * expansion_factor)
        self.activation = nn.SiLU()
        self.dropout1 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_model * expansion_factor, d_model)
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, x: Tensor) -> Tensor:
        h = self.dropout1(self.activation(_lin(_ln(x, self.layer_norm), self.linear1.weight, self.linear1.bias)))
        return self.dropout2(_lin(h, self.linear2.weight, self.linear2.bias))


class FastConformerBlock(nn.Module):
    """x + ff1/2 -> + attention -> + conv -> + ff2/2 -> LayerNorm (model.py:591-679)."""

    def __init__(self, d_model: int, num_heads: int, conv_kernel_size: int = 9, ff_expansion_factor: int = 4,
                 dropout: float = 0.1, use_local_attn: bool = False, local_context: int = 256,
                 stochastic_depth_prob: float = 0.0, conv_norm_type: str = "batch", use_causal_conv: bool = True):
        super().__init__()
        self.ff_scale = 0.5
        self.drop_prob = stochastic_depth_prob
        self.ff1 = FeedForwardModule(d_model, ff_expansion_factor, dropout)
        self.attention = RelPositionMultiHeadAttention(d_model=d_model, num_heads=num_heads, dropout=dropout,
                                                       use_local_attn=use_local_attn, local_context=local_context)
        self.conv = ConformerConvolution(d_model=d_model, kernel_size=conv_kernel_size, dropout=dropout,
                                         norm_type=conv_norm_type, use_causal=use_causal_conv)
        self.ff2 = FeedForwardModule(d_model, ff_expansion_factor, dropout)
        self.layer_norm = nn.LayerNorm(d_model)

    def forward(self, x: Tensor, pos_emb: Tensor, mask: Optional[Tensor] = None,
                attn_cache: Optional[Tensor] = None, conv_cache: Optional[Tensor] = None):
        if self.training and self.drop_prob > 0 and random.random() < self.drop_prob:
            return x, attn_cache, conv_cache            # stochastic depth: skip the block
        x = x + self.ff_scale * self.ff1(x)
        att, attn_cache = self.attention(x, pos_emb, mask, attn_cache)
        x = x + att
        cv, conv_cache = self.conv(x, conv_cache)
        x = x + cv
        x = x + self.ff_scale * self.ff2(x)
        return _ln(x, self.layer_norm), attn_cache, conv_cache


class Subsampling8x(nn.Module):
    """Three Conv2d(3x3, s2, p1)+BatchNorm2d+SiLU stages and a Linear over (channel, freq)
    (model.py:686-760)."""

    def __init__(self, d_input: int = 80, d_model: int = 256, hidden_channels: int = 256):
        super().__init__()

        def stage(cin):
            return nn.Sequential(nn.Conv2d(cin, hidden_channels, kernel_size=3, stride=2, padding=1),
                                 nn.BatchNorm2d(hidden_channels), nn.SiLU())

        self.conv1, self.conv2, self.conv3 = stage(1), stage(hidden_channels), stage(hidden_channels)
        self.d_input, self.d_model = d_input, d_model
        self.linear = nn.Linear(hidden_channels * self.get_output_length(d_input), d_model)

    def forward(self, x: Tensor) -> Tensor:
        y = self.conv3(self.conv2(self.conv1(x.unsqueeze(1))))
        return _lin(y.transpose(1, 2).flatten(2), self.linear.weight, self.linear.bias)

    @staticmethod
    def get_output_length(input_length: int) -> int:
        for _ in range(3):
            input_length = (input_length + 1) // 2
        return input_length


class FastConformerEncoder(nn.Module):
    """model.py:767-902.  `forward(x, lengths=None, cache=None, streaming=False)` ->
    (out (B,T',d) fp32, output_lengths int64 (B,), new_cache | None)."""

    def __init__(self, d_input: int = 80, d_model: int = 256, num_layers: int = 17, num_heads: int = 4,
                 conv_kernel_size: int = 9, ff_expansion_factor: int = 4, dropout: float = 0.1,
                 subsampling_channels: int = 256, use_local_attn: bool = False, local_context: int = 256,
                 stochastic_depth=None, conv_norm_type: str = "batch", use_causal_conv: bool = True,
                 compute_dtype: str = "fp32"):
        super().__init__()
        self.d_model, self.num_layers, self.use_causal_conv = d_model, num_layers, use_causal_conv
        self.subsampling = Subsampling8x(d_input, d_model, subsampling_channels)
        self.pos_encoding = RelativePositionalEncoding(d_model, dropout=dropout)
        if stochastic_depth is not None and stochastic_depth.enabled:
            probs = stochastic_depth_schedule(num_layers, stochastic_depth.drop_prob, stochastic_depth.mode,
                                              stochastic_depth.start_layer)
        else:
            probs = [0.0] * num_layers
        self.layers = nn.ModuleList([
            FastConformerBlock(d_model=d_model, num_heads=num_heads, conv_kernel_size=conv_kernel_size,
                               ff_expansion_factor=ff_expansion_factor, dropout=dropout,
                               use_local_attn=use_local_attn, local_context=local_context,
                               stochastic_depth_prob=probs[i], conv_norm_type=conv_norm_type,
                               use_causal_conv=use_causal_conv)
            for i in range(num_layers)])
        self.set_precision(compute_dtype)
        self._packed = {}
        self._packed_version = None
        self._weight_tensors = None
        self.stream_lanes = 0          # slices of a streaming batch run on concurrent HIP streams (0 = by batch size)
        self.forward_lanes = 0         # slices of an offline batch whose blocks run on concurrent HIP streams (0 = by batch size)
        # training-mode forwards on the GPU through the HIP kernels with autograd (conformer_amd.autograd): True, False,
        # or "auto" = True (measured, small config, forward + backward + AdamW, HIP path vs ATen: 16 x 10 s fp32 44 vs
        # 50 ms, AMP 48 vs 62 ms; 32 x 20 s fp32 70 vs 81 ms, AMP 50 vs 63 ms)
        self.native_training = "auto"
        self.stream_graph = False      # replay the streaming step from a captured HIP graph (PackedEncoder.stream_step)
        self.register_load_state_dict_post_hook(lambda module, _keys: module.invalidate_packed())

    # ---- native path plumbing -------------------------------------------------------------
    def set_precision(self, name: str) -> "FastConformerEncoder":
        """'fp32' (exact-fp32 MFMA, default like the reference) or 'bf16' (bf16 MFMA, fp32 accumulate)."""
        table = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}
        if name not in table:
            raise ValueError(f"compute_dtype must be one of {sorted(table)}, got {name!r}")
        self.compute_dtype = table[name]
        return self

    def invalidate_packed(self) -> None:
        """Drop packed weights.  load_state_dict, .to()/.cuda(), a switch to training mode and any in-place update
        of a parameter or buffer made through torch (optimizer steps, BatchNorm running statistics: train.py:336-355
        between two validate() calls, train.py:648) are noticed without this call, through the tensors' version
        counters.  Call it by hand after writes torch does not count: through `.data`, through a raw pointer or an
        external library, or to tensors created under torch.inference_mode() (they carry no version counter; for
        those only a change of storage is noticed)."""
        self._packed = {}                 # the workspaces belong to the packed encoders and go with them
        self._weight_tensors = None

    def train(self, mode: bool = True):
        # the loop of train.py:639-650 alternates model.train() / model.eval(): whatever was packed for the last
        # validation is stale once training resumes
        if mode:
            self.invalidate_packed()
        return super().train(mode)

    def _weights_version(self):
        """In-place writes bump a tensor's version counter: the tuple over every parameter and buffer changes
        whenever any of them was updated since it was last taken (about 0.1 ms on the host for 670 tensors)."""
        if self._weight_tensors is None:
            self._weight_tensors = list(self.parameters()) + list(self.buffers())
        try:
            return tuple(t._version for t in self._weight_tensors)
        except RuntimeError:
            # inference tensors do not track versions: fall back to where each tensor's storage lives (a re-assigned
            # or re-loaded tensor is noticed, an in-place write is not: invalidate_packed() covers that case)
            return tuple(t._version if not t.is_inference() else ("ptr", t.data_ptr()) for t in self._weight_tensors)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_packed()
        return super()._apply(fn, *args, **kwargs)

    def packed(self, device: torch.device, dtype: Optional[torch.dtype] = None, check: bool = True):
        """Packed weights for (device, dtype), re-packed when any parameter or buffer has been written since.
        check=False skips that comparison (the chunks of a running stream after the first)."""
        from .pack import PackedEncoder
        dtype = dtype or self.compute_dtype
        key = (str(device), dtype)
        if check and self._packed:
            version = self._weights_version()
            if version != self._packed_version:
                self.invalidate_packed()
        if key not in self._packed:
            version = self._weights_version()
            with torch.no_grad():
                self._packed[key] = PackedEncoder(self, dtype, device)
            self._packed_version = version
        return self._packed[key]

    def _native_dtype(self) -> torch.dtype:
        if torch.is_autocast_enabled():          # AMP callers (train.py:301) get the bf16 kernels
            return torch.bfloat16
        return self.compute_dtype

    # ---- forward --------------------------------------------------------------------------
    def forward(self, x: Tensor, lengths: Optional[Tensor] = None, cache: Optional[Cache] = None,
                streaming: bool = False):
        if self._native_streaming(x, lengths, cache, streaming):
            return self.forward_stream(x, cache)
        if self.training or streaming or cache is not None:
            if not self.training and x.is_cuda:
                # never silent: an eval-mode GPU call that the HIP step is not built for is reported
                warnings.warn("conformer_amd: this streaming call (lengths given, more than min(16, local_context) output "
                              "frames per chunk, or a local_context that is not a multiple of 8 up to 512) is outside what the HIP "
                              "streaming step is built for and runs the PyTorch composite instead", RuntimeWarning,
                              stacklevel=2)
            if isinstance(cache, StreamCache):
                cache = cache.to_reference()
            return self.forward_composite(x, lengths, cache, streaming)
        if not x.is_cuda:
            raise RuntimeError("conformer_amd: the eval-mode encoder runs on MI355X through libfcenc_hip.so only; "
                               "move the model and inputs to the GPU (there is no CPU execution path)")
        other = {layer.conv.norm_type for layer in self.layers} - {"batch"}
        if other:
            # model.py:484-493, 531-536: layer / instance / group / identity norms in the conv module.  The HIP conv
            # module folds an eval-mode BatchNorm into the depthwise conv; the other norms depend on the activations
            # and run op by op on ATen.  Never silent, and no parity or performance claim is made through it.
            warnings.warn(f"conformer_amd: conv_norm_type {sorted(other)} is outside what the HIP conv module is built for "
                          "(it folds BatchNorm); this forward runs the PyTorch composite instead", RuntimeWarning,
                          stacklevel=2)
            return self.forward_composite(x, lengths, cache, streaming)
        from . import ops                                   # registers the fcenc:: operators
        # one operator = one C call for the whole pass.  Under FakeTensorMode / torch.compile tracing only its shape
        # function runs: nothing is packed and the handle is never looked up.
        handle = 0 if _is_fake(x) else self.packed(x.device, self._native_dtype()).handle
        out, out_len = torch.ops.fcenc.encoder_forward(x.float().contiguous(), lengths, handle, self.d_model,
                                                       self.forward_lanes)
        return out, out_len, None

    def _native_streaming(self, x: Tensor, lengths, cache, streaming: bool) -> bool:
        """The cached chunk step runs on the HIP kernels in eval mode on the GPU, without lengths (the
        reference's streaming caller passes none, inference.py:358) and for chunk / cache sizes the
        kernels are built for; anything else takes the PyTorch composite."""
        if self.training or not streaming or lengths is not None or not x.is_cuda or x.dim() != 3:
            return False
        if not self.use_causal_conv:
            return False
        try:
            # a running stream keeps the weights it started with: only its first chunk pays for the comparison
            pk = self.packed(x.device, self._native_dtype(), check=not isinstance(cache, StreamCache))
        except NotImplementedError:
            return False
        if isinstance(cache, StreamCache) and (cache.packed is not pk or cache.batch != x.size(0)):
            return False
        return pk.stream_supported(Subsampling8x.get_output_length(x.size(1)))

    def forward_stream(self, x: Tensor, cache=None):
        """One chunk for every stream of the batch on the HIP kernels.  `cache`: None (start of the streams),
        the StreamCache returned by the previous chunk, or a cache in the reference's format (imported)."""
        pk = self.packed(x.device, self._native_dtype(), check=False)
        bsz = x.size(0)
        if cache is None:
            cache = pk.new_stream_cache(bsz)
        elif not isinstance(cache, StreamCache):
            cache = StreamCache.from_reference(pk, cache, bsz)
        from . import ops                                   # registers the fcenc:: operators
        out = torch.ops.fcenc.encoder_stream_step(x.float().contiguous(), cache.k, cache.v, cache.conv, pk.handle,
                                                  cache.handle, self.d_model, self.stream_lanes, self.stream_graph)
        out_len = torch.full((bsz,), out.size(1), device=x.device)
        return out, out_len, cache

    def forward_composite(self, x: Tensor, lengths: Optional[Tensor] = None, cache: Optional[Cache] = None,
                          streaming: bool = False):
        """The module tree run op by op with autograd, dropout, batch statistics and per-chunk caches (training and
        the streaming calls the HIP step is not built for).  In training mode on the GPU the matrix products,
        LayerNorms and the GLU + depthwise core run on the HIP kernels (conformer_amd.autograd); everything else,
        and every CPU call, is ATen."""
        want = True if self.native_training == "auto" else bool(self.native_training)
        native = self.training and want and x.is_cuda and not _is_fake(x)
        prev = (_native_on(), getattr(_native, "bf16", False), getattr(_native, "streaming", False))
        _native.on, _native.streaming = native, streaming
        _native.bf16 = native and torch.is_autocast_enabled()        # AMP training (train.py:301): bf16 operands, fp32 accumulate
        try:
            return self._forward_composite(x, lengths, cache, streaming)
        finally:
            _native.on, _native.bf16, _native.streaming = prev

    def _forward_composite(self, x: Tensor, lengths: Optional[Tensor], cache: Optional[Cache], streaming: bool):
        x = torch.log(x + 1e-6)
        x = (x - x.mean(dim=(1, 2), keepdim=True)) / (x.std(dim=(1, 2), keepdim=True) + 1e-6)
        x = self.subsampling(x)
        bsz, t_out = x.size(0), x.size(1)
        mask = None
        if lengths is None:
            out_len = torch.full((bsz,), t_out, device=x.device)
        else:
            out_len = torch.tensor([Subsampling8x.get_output_length(int(n)) for n in lengths.tolist()],
                                   device=x.device)
            mask = torch.arange(t_out, device=x.device).unsqueeze(0) >= out_len.unsqueeze(1)
        x, pos_emb = self.pos_encoding(x)
        cache = cache if cache is not None else [(None, None)] * self.num_layers
        new_cache = []
        for layer, (ac, cc) in zip(self.layers, cache):
            x, ac, cc = layer(x, pos_emb, mask, ac, cc)
            new_cache.append((ac, cc))
        return x, out_len, (new_cache if streaming else None)

    def get_initial_cache(self, batch_size: int, device: torch.device) -> Cache:
        return [(None, layer.conv.get_initial_cache(batch_size, device)) for layer in self.layers]
