"""Deterministic synthetic weights and inputs (no datasets or checkpoints are reachable).

`make_state_dict` builds a complete `model_state_dict` with the reference's key names and
shapes (model.py:1015-1092 module tree) straight from a config dict and a seed, without
touching any nn.Module initialiser, so the same tensors can be recreated bit-for-bit on
any box from (config, seed).  BatchNorm running statistics are drawn away from (0, 1) so
that BatchNorm folding is exercised.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch


def _freq_out(d_input: int) -> int:
    f = d_input
    for _ in range(3):
        f = (f + 1) // 2
    return f


def state_dict_shapes(config: dict) -> Dict[str, Tuple[int, ...]]:
    """Key -> shape of the reference model's state_dict for `config` (insertion order =
    the reference's registration order)."""
    mc = config.get("model", {})
    d_in = mc.get("d_input", 80)
    vocab = mc.get("vocab_size", 29)
    d = mc.get("d_encoder", 256)
    dp = mc.get("d_predictor", 256)
    dj = mc.get("d_joint", 320)
    nl = mc.get("encoder_layers", 17)
    h = mc.get("encoder_heads", 4)
    k = mc.get("encoder_conv_kernel", 9)
    ff = mc.get("encoder_ff_expansion", 4) * d
    pl = mc.get("predictor_layers", 2)
    pk = mc.get("predictor_kernel", 2)
    c = mc.get("subsampling_channels", 256)
    norm = mc.get("conv_norm_type", "batch")
    s: Dict[str, Tuple[int, ...]] = {}
    cin = 1
    for i in (1, 2, 3):
        p = f"encoder.subsampling.conv{i}"
        s[p + ".0.weight"] = (c, cin, 3, 3)
        s[p + ".0.bias"] = (c,)
        for nm in ("weight", "bias", "running_mean", "running_var"):
            s[f"{p}.1.{nm}"] = (c,)
        s[p + ".1.num_batches_tracked"] = ()
        cin = c
    s["encoder.subsampling.linear.weight"] = (d, c * _freq_out(d_in))
    s["encoder.subsampling.linear.bias"] = (d,)
    s["encoder.pos_encoding.pe"] = (2 * 5000 - 1, d)
    for i in range(nl):
        p = f"encoder.layers.{i}"
        for ffn in ("ff1",):
            s[f"{p}.{ffn}.layer_norm.weight"] = (d,)
            s[f"{p}.{ffn}.layer_norm.bias"] = (d,)
            s[f"{p}.{ffn}.linear1.weight"] = (ff, d)
            s[f"{p}.{ffn}.linear1.bias"] = (ff,)
            s[f"{p}.{ffn}.linear2.weight"] = (d, ff)
            s[f"{p}.{ffn}.linear2.bias"] = (d,)
        s[p + ".attention.pos_bias_u"] = (h, d // h)
        s[p + ".attention.pos_bias_v"] = (h, d // h)
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[f"{p}.attention.{nm}.weight"] = (d, d)
            s[f"{p}.attention.{nm}.bias"] = (d,)
        s[p + ".attention.pos_proj.weight"] = (d, d)
        s[p + ".conv.layer_norm.weight"] = (d,)
        s[p + ".conv.layer_norm.bias"] = (d,)
        s[p + ".conv.pointwise1.weight"] = (2 * d, d, 1)
        s[p + ".conv.pointwise1.bias"] = (2 * d,)
        s[p + ".conv.depthwise.conv.weight"] = (d, 1, k)
        s[p + ".conv.depthwise.conv.bias"] = (d,)
        if norm == "batch":
            for nm in ("weight", "bias", "running_mean", "running_var"):
                s[f"{p}.conv.norm.{nm}"] = (d,)
            s[p + ".conv.norm.num_batches_tracked"] = ()
        elif norm in ("layer", "group"):
            s[p + ".conv.norm.weight"] = (d,)
            s[p + ".conv.norm.bias"] = (d,)
        s[p + ".conv.pointwise2.weight"] = (d, d, 1)
        s[p + ".conv.pointwise2.bias"] = (d,)
        s[f"{p}.ff2.layer_norm.weight"] = (d,)
        s[f"{p}.ff2.layer_norm.bias"] = (d,)
        s[f"{p}.ff2.linear1.weight"] = (ff, d)
        s[f"{p}.ff2.linear1.bias"] = (ff,)
        s[f"{p}.ff2.linear2.weight"] = (d, ff)
        s[f"{p}.ff2.linear2.bias"] = (d,)
        s[p + ".layer_norm.weight"] = (d,)
        s[p + ".layer_norm.bias"] = (d,)
    s["predictor.embedding.weight"] = (vocab, dp)
    for i in range(pl):
        s[f"predictor.convs.{i}.conv.weight"] = (dp, dp, pk)
        s[f"predictor.convs.{i}.conv.bias"] = (dp,)
    s["predictor.layer_norm.weight"] = (dp,)
    s["predictor.layer_norm.bias"] = (dp,)
    s["joint.encoder_proj.weight"] = (dj, d)
    s["joint.encoder_proj.bias"] = (dj,)
    s["joint.predictor_proj.weight"] = (dj, dp)
    s["joint.predictor_proj.bias"] = (dj,)
    s["joint.output_proj.weight"] = (vocab, dj)
    s["joint.output_proj.bias"] = (vocab,)
    return s


def sinusoid_table(d_model: int, max_len: int = 5000) -> torch.Tensor:
    """The relative sinusoid table: row p+(max_len-1) holds position p in
    [-(max_len-1), max_len-1]; even columns sin(p*w_i), odd columns cos(p*w_i),
    w_i = exp(-2i ln(1e4)/d)   (model.py:134-146)."""
    pos = torch.arange(-(max_len - 1), max_len, dtype=torch.float32)
    dim = torch.arange(0, d_model, 2, dtype=torch.float32)
    div = torch.exp(dim * (-math.log(10000.0) / d_model))
    pe = torch.zeros(2 * max_len - 1, d_model)
    pe[:, 0::2] = torch.sin(pos.unsqueeze(1) * div)
    pe[:, 1::2] = torch.cos(pos.unsqueeze(1) * div)
    return pe


def _draw(key: str, shape: Tuple[int, ...], g: torch.Generator, d: int) -> torch.Tensor:
    leaf = key.rsplit(".", 1)[-1]
    if key.endswith("pos_encoding.pe"):
        return sinusoid_table(d)
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.int64)
    if leaf == "running_mean":
        return torch.randn(shape, generator=g) * 0.1
    if leaf == "running_var":
        return torch.rand(shape, generator=g) + 0.5
    if "norm" in key and leaf == "weight" or (".1.weight" in key and "subsampling" in key):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf == "bias":
        return 0.05 * torch.randn(shape, generator=g)
    if "pos_bias_" in key:
        return torch.randn(shape, generator=g) * 0.1
    if key.endswith("embedding.weight"):
        return torch.randn(shape, generator=g)
    fan_in = 1
    for n in shape[1:]:
        fan_in *= n
    return torch.randn(shape, generator=g) / math.sqrt(max(fan_in, 1))


def make_state_dict(config: dict, seed: int, decoder_seed: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """All tensors from one generator seeded with `seed`; with `decoder_seed` the predictor.* and joint.* tensors come
    from a second generator instead, so that the decoder can be re-drawn without changing the encoder."""
    g_enc = torch.Generator().manual_seed(seed)
    g_dec = g_enc if decoder_seed is None else torch.Generator().manual_seed(decoder_seed)
    d = config.get("model", {}).get("d_encoder", 256)
    return {key: _draw(key, shape, g_enc if key.startswith("encoder.") else g_dec, d)
            for key, shape in state_dict_shapes(config).items()}


def decoder_state_dict(config: dict, decoder_seed: int) -> Dict[str, torch.Tensor]:
    """Only the predictor.* and joint.* tensors of `make_state_dict(config, any, decoder_seed)`."""
    g = torch.Generator().manual_seed(decoder_seed)
    d = config.get("model", {}).get("d_encoder", 256)
    return {key: _draw(key, shape, g, d) for key, shape in state_dict_shapes(config).items()
            if not key.startswith("encoder.")}


def synthetic_mel(batch: int, frames: int, lengths: Optional[List[int]], seed: int,
                  n_mels: int = 80) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Strictly positive power-mel `rand*10 + 1e-3` (the encoder applies log itself,
    model.py:856), zeroed beyond each utterance's length as train.py:174 pads."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, frames, n_mels, generator=g) * 10 + 1e-3
    if lengths is None:
        return x, None
    ln = torch.tensor(lengths, dtype=torch.int64)
    x = x * (torch.arange(frames)[None, :, None] < ln[:, None, None])
    return x, ln


def synthetic_lengths(batch: int, frames: int, seed: int, min_frac: float = 0.5,
                      min_frames: int = 0) -> List[int]:
    """max == frames, the rest uniform in [max(min_frac*frames, min_frames), frames]."""
    g = torch.Generator().manual_seed(seed + 7919)
    lo = max(int(math.ceil(min_frac * frames)), min_frames)
    ln = torch.randint(lo, frames + 1, (batch,), generator=g).tolist()
    ln[0] = frames
    return ln
