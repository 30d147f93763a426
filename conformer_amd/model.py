"""`create_model(config)` and the FastConformerTransducer shell -- the module API callers of the
reference use (`train.py:302, :233, :528`, `inference.py:139, :358, :548-554`), with the
encoder replaced by the MI355X-native one in `modules.py`.  The predictor and joint networks
keep their PyTorch definitions (model.py:909-1008): they are outside the accelerated path and
only feed the greedy / beam decode loops.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .modules import CausalConv1d, FastConformerEncoder

Tensor = torch.Tensor


@dataclass
class StochasticDepthConfig:
    enabled: bool = True
    drop_prob: float = 0.1
    mode: str = "linear"
    start_layer: int = 1


class StatelessPredictor(nn.Module):
    """Embedding followed by residual causal Conv1d layers and a LayerNorm; no recurrent state."""

    def __init__(self, vocab_size: int, d_model: int = 256, num_layers: int = 2, kernel_size: int = 2,
                 dropout: float = 0.1):
        super().__init__()
        self.embedding = nn.Embedding(vocab_size, d_model)
        self.embed_dropout = nn.Dropout(dropout)
        self.convs = nn.ModuleList([CausalConv1d(d_model, d_model, kernel_size) for _ in range(num_layers)])
        self.activation = nn.SiLU()
        self.layer_norm = nn.LayerNorm(d_model)

    def forward(self, y: Tensor) -> Tensor:
        x = self.embed_dropout(self.embedding(y)).transpose(1, 2)
        for conv in self.convs:
            x = self.activation(conv(x, None)[0]) + x
        return self.layer_norm(x.transpose(1, 2))


class JointNetwork(nn.Module):
    """logits[b,t,u] = W_o tanh(W_e enc[b,t] + W_p pred[b,u]), clamped to +-logit_clamp."""

    def __init__(self, d_encoder: int, d_predictor: int, d_joint: int, vocab_size: int, logit_clamp: float = 15.0):
        super().__init__()
        self.encoder_proj = nn.Linear(d_encoder, d_joint)
        self.predictor_proj = nn.Linear(d_predictor, d_joint)
        self.output_proj = nn.Linear(d_joint, vocab_size)
        self.logit_clamp = logit_clamp

    def forward(self, encoder_out: Tensor, predictor_out: Tensor) -> Tensor:
        h = torch.tanh(self.encoder_proj(encoder_out).unsqueeze(2) + self.predictor_proj(predictor_out).unsqueeze(1))
        logits = self.output_proj(h)
        return logits.clamp(-self.logit_clamp, self.logit_clamp) if self.logit_clamp > 0 else logits


class FastConformerTransducer(nn.Module):
    def __init__(self, d_input: int = 80, vocab_size: int = 29, d_encoder: int = 256, d_predictor: int = 256,
                 d_joint: int = 320, encoder_layers: int = 17, encoder_heads: int = 4, encoder_conv_kernel: int = 9,
                 encoder_ff_expansion: int = 4, predictor_layers: int = 2, predictor_kernel: int = 2,
                 dropout: float = 0.1, subsampling_channels: int = 256, use_local_attn: bool = True,
                 local_context: int = 256, stochastic_depth_prob: float = 0.1, conv_norm_type: str = "batch",
                 compute_dtype: str = "fp32"):
        super().__init__()
        self.vocab_size, self.blank_id, self.d_encoder = vocab_size, 0, d_encoder
        depth = StochasticDepthConfig(enabled=stochastic_depth_prob > 0, drop_prob=stochastic_depth_prob,
                                      mode="linear", start_layer=1)
        self.encoder = FastConformerEncoder(
            d_input=d_input, d_model=d_encoder, num_layers=encoder_layers, num_heads=encoder_heads,
            conv_kernel_size=encoder_conv_kernel, ff_expansion_factor=encoder_ff_expansion, dropout=dropout,
            subsampling_channels=subsampling_channels, use_local_attn=use_local_attn, local_context=local_context,
            stochastic_depth=depth, conv_norm_type=conv_norm_type, use_causal_conv=True,
            compute_dtype=compute_dtype)
        self.predictor = StatelessPredictor(vocab_size=vocab_size, d_model=d_predictor, num_layers=predictor_layers,
                                            kernel_size=predictor_kernel, dropout=dropout)
        self.joint = JointNetwork(d_encoder=d_encoder, d_predictor=d_predictor, d_joint=d_joint,
                                  vocab_size=vocab_size)

    def forward(self, audio: Tensor, audio_lengths: Tensor, targets: Tensor, target_lengths: Tensor,
                debug: bool = False) -> Tuple[Tensor, Tensor]:
        """Training forward: logits (B, T', U+1, V) and encoder lengths."""
        enc, enc_len, _ = self.encoder(audio, audio_lengths)
        start = targets.new_zeros((targets.size(0), 1))
        pred = self.predictor(torch.cat([start, targets], dim=1))
        logits = self.joint(enc, pred)
        if debug:
            for name, t in (("input.audio", audio), ("encoder.output", enc), ("predictor.output", pred),
                            ("joint.logits", logits)):
                f = t.detach().float()
                print(f"[DEBUG] {name:20s} shape={list(t.shape)} min={f.min().item():+.4e} max={f.max().item():+.4e} "
                      f"mean={f.mean().item():+.4e} finite={bool(torch.isfinite(f).all())}")
        return logits, enc_len

    def encode(self, audio: Tensor, audio_lengths: Optional[Tensor] = None, cache=None, streaming: bool = False):
        return self.encoder(audio, audio_lengths, cache, streaming)

    def predict(self, targets: Tensor) -> Tensor:
        return self.predictor(targets)

    def joint_step(self, encoder_out: Tensor, predictor_out: Tensor) -> Tensor:
        return self.joint(encoder_out, predictor_out).squeeze(1).squeeze(1)

    def get_initial_cache(self, batch_size: int, device: torch.device):
        return self.encoder.get_initial_cache(batch_size, device)


_MODEL_KEYS = dict(d_input=80, vocab_size=29, d_encoder=256, d_predictor=256, d_joint=320, encoder_layers=17,
                   encoder_heads=4, encoder_conv_kernel=9, encoder_ff_expansion=4, predictor_layers=2,
                   predictor_kernel=2, dropout=0.1, subsampling_channels=256, use_local_attn=True,
                   local_context=256, stochastic_depth_prob=0.1, conv_norm_type="batch")


def create_model(config: dict) -> FastConformerTransducer:
    """Same keys and defaults as the reference's create_model (model.py:1213-1235); the extra
    optional key `compute_dtype` ("fp32" | "bf16") selects the kernel precision."""
    mc = config.get("model", {})
    kwargs = {k: mc.get(k, default) for k, default in _MODEL_KEYS.items()}
    return FastConformerTransducer(compute_dtype=mc.get("compute_dtype", "fp32"), **kwargs)


def load_config(path: str) -> dict:
    with open(path, "r") as fh:
        return json.load(fh)


def model_size(model: nn.Module, name: str = "Model") -> dict:
    n = sum(p.numel() for p in model.parameters())
    nbytes = sum(t.numel() * t.element_size() for t in list(model.parameters()) + list(model.buffers()))
    print(f"{name}: {n / 1e6:.2f}M parameters, {nbytes / 1024 ** 2:.2f} MB")
    return {"num_params": n, "size_mb": nbytes / 1024 ** 2}
