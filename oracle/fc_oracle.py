"""CPU oracle for the FastConformer encoder forward pass -- TEST INFRASTRUCTURE ONLY.

This file restates, in plain functional fp32 PyTorch on CPU, the arithmetic of the
reference encoder (`/root/reference/model.py`).  It is the checker the HIP path is
compared against; nothing under ``conformer_amd/`` imports it and it is never the
thing that is measured or shipped.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may call it.

Pinning: the reference holds no golden vectors or tests of its own, so the oracle is
pinned by outputs of the reference itself: ``tests/golden/gen_golden.py`` imports
``/root/reference/model.py`` in the build container, runs it on seeded inputs and
commits the inputs + outputs as fixtures; ``tests/test_oracle_golden.py`` checks this
file against them (<= 1e-5) and, where the reference is mounted, against the live
reference for the full small / medium / large configs.

Every function takes a flat ``state_dict`` (the checkpoint's ``model_state_dict``) and
reads the tensors by the reference's key names, so it runs on any checkpoint the
reference can load.  Each function cites the reference lines it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

BN_EPS = 1e-5   # nn.BatchNorm default, model.py:705,485
LN_EPS = 1e-5   # nn.LayerNorm default, model.py:465,570,635


# ----------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------
def encoder_hparams(config: dict) -> dict:
    """Defaults of `create_model` (model.py:1213-1235) restricted to the encoder."""
    mc = config.get("model", {})
    return dict(
        d_input=mc.get("d_input", 80),
        d_model=mc.get("d_encoder", 256),
        num_layers=mc.get("encoder_layers", 17),
        num_heads=mc.get("encoder_heads", 4),
        conv_kernel=mc.get("encoder_conv_kernel", 9),
        ff_expansion=mc.get("encoder_ff_expansion", 4),
        channels=mc.get("subsampling_channels", 256),
        use_local_attn=mc.get("use_local_attn", True),
        local_context=mc.get("local_context", 256),
        conv_norm_type=mc.get("conv_norm_type", "batch"),
    )


# ----------------------------------------------------------------------------
# integer work (bit-exact)
# ----------------------------------------------------------------------------
def subsampled_length(n: int) -> int:
    """model.py:755-760: three rounds of ceil(n/2)."""
    for _ in range(3):
        n = (n + 1) // 2
    return n


def output_lengths_and_mask(lengths: Optional[Tensor], batch: int, t_out: int
                            ) -> Tuple[Tensor, Optional[Tensor]]:
    """model.py:863-876. Returns (int64 (B,), bool (B,T') with True = padding) or mask None."""
    if lengths is None:
        return torch.full((batch,), t_out, dtype=torch.int64), None
    out = torch.tensor([subsampled_length(int(v)) for v in lengths.tolist()], dtype=torch.int64)
    mask = torch.arange(t_out).unsqueeze(0) >= out.unsqueeze(1)
    return out, mask


# ----------------------------------------------------------------------------
# stages
# ----------------------------------------------------------------------------
def input_norm(x: Tensor) -> Tensor:
    """model.py:856-857: log compression, per-utterance standardisation over (time, freq)
    with the unbiased std, padding frames included."""
    x = torch.log(x + 1e-6)
    mu = x.mean(dim=(1, 2), keepdim=True)
    sd = x.std(dim=(1, 2), keepdim=True)
    return (x - mu) / (sd + 1e-6)


def _bn_eval(x: Tensor, sd: SD, prefix: str, channel_dim: int) -> Tensor:
    """BatchNorm in eval mode: (x - running_mean) / sqrt(running_var + eps) * weight + bias."""
    assert channel_dim == 1
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, BN_EPS)


def subsampling_convs(x: Tensor, sd: SD, prefix: str = "encoder.subsampling") -> List[Tensor]:
    """model.py:737-743: three times Conv2d(3x3, stride 2, pad 1) + BatchNorm2d(eval) + SiLU.
    x: (B,T,F) normalised. Returns the three NCHW activations."""
    outs = []
    h = x.unsqueeze(1)
    for i in (1, 2, 3):
        h = F.conv2d(h, sd[f"{prefix}.conv{i}.0.weight"], sd[f"{prefix}.conv{i}.0.bias"],
                     stride=2, padding=1)
        h = F.silu(_bn_eval(h, sd, f"{prefix}.conv{i}.1", 1))
        outs.append(h)
    return outs


def subsampling_linear(h3: Tensor, sd: SD, prefix: str = "encoder.subsampling") -> Tensor:
    """model.py:746-751: (B,C,T',F') -> (B,T',C*F') with index c*F'+f, then Linear."""
    b, c, t, f = h3.shape
    flat = h3.permute(0, 2, 1, 3).reshape(b, t, c * f)
    return F.linear(flat, sd[f"{prefix}.linear.weight"], sd[f"{prefix}.linear.bias"])


def rel_pos_table(sd: SD, t_out: int, d_model: int) -> Tensor:
    """model.py:175-180: rows (max_len-1)-T'+1 .. (max_len-1)+T'-1 of the `pe` buffer;
    row r of the result is relative position r-(T'-1).  The buffer is rebuilt from the
    closed form (model.py:134-146) when T' exceeds its length (model.py:154-157)."""
    pe = sd["encoder.pos_encoding.pe"]
    max_len = (pe.shape[0] + 1) // 2
    if t_out > max_len:
        max_len = max(t_out, max_len * 2)
        pos = torch.arange(-(max_len - 1), max_len, dtype=torch.float32)
        dim = torch.arange(0, d_model, 2, dtype=torch.float32)
        div = torch.exp(dim * (-math.log(10000.0) / d_model))
        pe = torch.zeros(2 * max_len - 1, d_model)
        pe[:, 0::2] = torch.sin(pos.unsqueeze(1) * div)
        pe[:, 1::2] = torch.cos(pos.unsqueeze(1) * div)
    centre = max_len - 1
    return pe[centre - t_out + 1: centre + t_out]


def feed_forward(x: Tensor, sd: SD, p: str) -> Tensor:
    """model.py:577-584 (dropout is the identity in eval)."""
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], LN_EPS)
    h = F.silu(F.linear(h, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"]))
    return F.linear(h, sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])


def shifted_pos_score(bd: Tensor, kv_len: int) -> Tensor:
    """model.py:312-325 followed by the size fix-up at model.py:394-400.

    bd: (B,H,Tq,2Tq-1).  Closed form of the pad/view/slice trick: the output element
    (i, jp) is element f = i*(2Tq-1) + jp + Tq of the row-major (Tq, 2Tq) left-padded
    matrix, i.e. 0 where f % 2Tq == 0, else bd[f // 2Tq, f % 2Tq - 1].  For Tq == kv_len
    this is bd[i, Tq-1+j-i].  The result is cut to the first kv_len columns, or left-padded
    with zeros up to kv_len."""
    b, h, tq, w = bd.shape
    f = (torch.arange(tq).unsqueeze(1) * w + torch.arange(w).unsqueeze(0) + tq)
    r = torch.div(f, 2 * tq, rounding_mode="floor")
    c = f % (2 * tq)
    gathered = bd[:, :, r.clamp(max=tq - 1), (c - 1).clamp(min=0)]
    out = torch.where((c == 0), torch.zeros((), dtype=bd.dtype), gathered)
    if w > kv_len:
        out = out[..., :kv_len]
    elif w < kv_len:
        out = F.pad(out, (kv_len - w, 0))
    return out


def local_window_mask(tq: int, kv_len: int, local_context: int) -> Tensor:
    """model.py:406-411: True = masked. Query i may see keys
    [max(0, kv-tq+i-lc+1), kv-tq+i]."""
    i = torch.arange(tq).unsqueeze(1)
    j = torch.arange(kv_len).unsqueeze(0)
    end = kv_len - tq + i + 1
    start = (end - local_context).clamp(min=0)
    return ~((j >= start) & (j < end))


def rel_pos_attention(x: Tensor, pos_emb: Tensor, mask: Optional[Tensor], cache: Optional[Tensor],
                      sd: SD, p: str, num_heads: int, use_local_attn: bool, local_context: int
                      ) -> Tuple[Tensor, Tensor]:
    """model.py:327-431. Returns (out_proj output, new kv cache (B,kv,2d))."""
    bsz, tq, d = x.shape
    dh = d // num_heads
    q = F.linear(x, sd[p + ".q_proj.weight"], sd[p + ".q_proj.bias"])
    k = F.linear(x, sd[p + ".k_proj.weight"], sd[p + ".k_proj.bias"])
    v = F.linear(x, sd[p + ".v_proj.weight"], sd[p + ".v_proj.bias"])
    if cache is not None:                                   # model.py:354-364
        ck, cv = cache.chunk(2, dim=-1)
        k = torch.cat([ck, k], dim=1)
        v = torch.cat([cv, v], dim=1)
        if k.shape[1] > local_context:
            k = k[:, -local_context:]
            v = v[:, -local_context:]
    new_cache = torch.cat([k, v], dim=-1)
    kv = k.shape[1]
    qh = q.view(bsz, tq, num_heads, dh)
    kh = k.view(bsz, kv, num_heads, dh).permute(0, 2, 1, 3)          # (B,H,kv,dh)
    vh = v.view(bsz, kv, num_heads, dh).permute(0, 2, 1, 3)
    ph = F.linear(pos_emb, sd[p + ".pos_proj.weight"]).view(-1, num_heads, dh).permute(1, 0, 2)
    qu = (qh + sd[p + ".pos_bias_u"]).permute(0, 2, 1, 3)            # (B,H,Tq,dh)
    qv = (qh + sd[p + ".pos_bias_v"]).permute(0, 2, 1, 3)
    ac = torch.matmul(qu, kh.transpose(-1, -2))                       # (B,H,Tq,kv)
    bd = torch.matmul(qv, ph.transpose(-1, -2).unsqueeze(0))          # (B,H,Tq,2Tq-1)
    score = (ac + shifted_pos_score(bd, kv)) * (dh ** -0.5)
    if use_local_attn and kv > local_context:
        score = score.masked_fill(local_window_mask(tq, kv, local_context), float("-inf"))
    if mask is not None:
        score = score.masked_fill(mask[:, None, None, :], float("-inf"))
    w = torch.softmax(score, dim=-1)
    o = torch.matmul(w, vh).permute(0, 2, 1, 3).reshape(bsz, tq, d)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"]), new_cache


def conv_module(x: Tensor, cache: Optional[Tensor], sd: SD, p: str, kernel: int,
                norm_type: str = "batch") -> Tuple[Tensor, Tensor, Dict[str, Tensor]]:
    """model.py:502-544 with the causal depthwise conv of model.py:222-249.
    Returns (output (B,T,d), new conv cache (B,d,k-1), intermediates)."""
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], LN_EPS)
    h = F.conv1d(h.transpose(1, 2), sd[p + ".pointwise1.weight"], sd[p + ".pointwise1.bias"])
    glu = h[:, :d] * torch.sigmoid(h[:, d:])                          # nn.GLU(dim=1)
    if cache is not None:
        padded = torch.cat([cache, glu], dim=-1)
    else:
        padded = F.pad(glu, (kernel - 1, 0))
    new_cache = padded[:, :, -(kernel - 1):].clone()
    dw = F.conv1d(padded, sd[p + ".depthwise.conv.weight"], sd[p + ".depthwise.conv.bias"], groups=d)
    if norm_type == "batch":
        n = _bn_eval(dw, sd, p + ".norm", 1)
    elif norm_type == "layer":
        n = F.layer_norm(dw.transpose(1, 2), (d,), sd[p + ".norm.weight"], sd[p + ".norm.bias"],
                         LN_EPS).transpose(1, 2)
    elif norm_type == "instance":
        n = F.instance_norm(dw)
    elif norm_type == "group":
        n = F.group_norm(dw, 8, sd[p + ".norm.weight"], sd[p + ".norm.bias"])
    else:
        n = dw
    act = F.silu(n)
    out = F.conv1d(act, sd[p + ".pointwise2.weight"], sd[p + ".pointwise2.bias"]).transpose(1, 2)
    return out, new_cache, {"glu": glu.transpose(1, 2), "dw_act": act.transpose(1, 2)}


def conformer_block(x: Tensor, pos_emb: Tensor, mask: Optional[Tensor],
                    attn_cache: Optional[Tensor], conv_cache: Optional[Tensor],
                    sd: SD, p: str, hp: dict, trace: Optional[dict] = None
                    ) -> Tuple[Tensor, Tensor, Tensor]:
    """model.py:662-679 (the training-only stochastic-depth skip at :659 is not taken in eval)."""
    x = x + 0.5 * feed_forward(x, sd, p + ".ff1")
    if trace is not None:
        trace["after_ff1"] = x
    a, new_attn = rel_pos_attention(x, pos_emb, mask, attn_cache, sd, p + ".attention",
                                    hp["num_heads"], hp["use_local_attn"], hp["local_context"])
    x = x + a
    if trace is not None:
        trace["after_attn"] = x
    c, new_conv, inter = conv_module(x, conv_cache, sd, p + ".conv", hp["conv_kernel"],
                                     hp["conv_norm_type"])
    x = x + c
    if trace is not None:
        trace["after_conv"] = x
        trace.update(inter)
    x = x + 0.5 * feed_forward(x, sd, p + ".ff2")
    if trace is not None:
        trace["after_ff2"] = x
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], LN_EPS)
    return x, new_attn, new_conv


@torch.no_grad()
def encoder_forward(sd: SD, config: dict, x: Tensor, lengths: Optional[Tensor] = None,
                    cache: Optional[List[Tuple[Optional[Tensor], Optional[Tensor]]]] = None,
                    streaming: bool = False, trace: Optional[dict] = None):
    """model.py:835-894.  Returns (out (B,T',d) fp32, out_lengths int64 (B,), cache list or None).
    When `trace` is a dict it is filled with the per-stage tensors used by the kernel tests."""
    hp = encoder_hparams(config)
    x = x.float()
    xn = input_norm(x)
    c1, c2, c3 = subsampling_convs(xn, sd)
    h = subsampling_linear(c3, sd)
    bsz, t_out, d = h.shape
    out_len, mask = output_lengths_and_mask(lengths, bsz, t_out)
    h = h * math.sqrt(d)                                              # model.py:172
    pos_emb = rel_pos_table(sd, t_out, d)
    if trace is not None:
        trace.update(dict(input_norm=xn, conv1=c1, conv2=c2, conv3=c3, embed=h,
                          pos_emb=pos_emb, mask=mask, layers=[]))
    if cache is None:
        cache = [(None, None)] * hp["num_layers"]
    new_cache = []
    for i in range(hp["num_layers"]):
        lt = {} if trace is not None else None
        h, na, nc = conformer_block(h, pos_emb, mask, cache[i][0], cache[i][1], sd,
                                    f"encoder.layers.{i}", hp, lt)
        if trace is not None:
            lt["out"] = h
            trace["layers"].append(lt)
        new_cache.append((na, nc))
    return h, out_len, (new_cache if streaming else None)


# ----------------------------------------------------------------------------
# unchanged predictor / joint + greedy decode (for the token-identity check)
# ----------------------------------------------------------------------------
def predictor_forward(sd: SD, y: Tensor, num_layers: int, kernel: int) -> Tensor:
    """model.py:939-961."""
    x = F.embedding(y, sd["predictor.embedding.weight"]).transpose(1, 2)
    for i in range(num_layers):
        res = x
        x = F.conv1d(F.pad(x, (kernel - 1, 0)), sd[f"predictor.convs.{i}.conv.weight"],
                     sd[f"predictor.convs.{i}.conv.bias"])
        x = F.silu(x) + res
    x = x.transpose(1, 2)
    d = x.shape[-1]
    return F.layer_norm(x, (d,), sd["predictor.layer_norm.weight"], sd["predictor.layer_norm.bias"], LN_EPS)


def joint_step(sd: SD, enc_t: Tensor, pred: Tensor, clamp: float = 15.0) -> Tensor:
    """model.py:986-1008 and :1191-1198 for a single (frame, symbol) pair -> (1,V)."""
    e = F.linear(enc_t, sd["joint.encoder_proj.weight"], sd["joint.encoder_proj.bias"]).unsqueeze(2)
    p = F.linear(pred, sd["joint.predictor_proj.weight"], sd["joint.predictor_proj.bias"]).unsqueeze(1)
    logits = F.linear(torch.tanh(e + p), sd["joint.output_proj.weight"], sd["joint.output_proj.bias"])
    return logits.clamp(-clamp, clamp).squeeze(1).squeeze(1)


@torch.no_grad()
def greedy_decode(sd: SD, config: dict, enc: Tensor, max_symbols: int = 10
                  ) -> Tuple[List[List[int]], float]:
    """train.py:180-211 / inference.py:161-194: per utterance, for every frame (padded frames
    included) emit argmax symbols until blank (0) or `max_symbols`; the predictor sees only the
    last emitted token.  Also returns the smallest top-1/top-2 logit gap over all decisions."""
    mc = config.get("model", {})
    nl, ks = mc.get("predictor_layers", 2), mc.get("predictor_kernel", 2)
    cache: Dict[int, Tensor] = {}

    def pred_of(tok: int) -> Tensor:
        if tok not in cache:
            cache[tok] = predictor_forward(sd, torch.tensor([[tok]]), nl, ks)
        return cache[tok]

    decoded, min_gap = [], float("inf")
    for b in range(enc.shape[0]):
        toks: List[int] = []
        po = pred_of(0)
        for t in range(enc.shape[1]):
            e = enc[b:b + 1, t:t + 1].float()
            for _ in range(max_symbols):
                logits = joint_step(sd, e, po)
                top = logits.topk(2, dim=-1).values[0]
                min_gap = min(min_gap, float(top[0] - top[1]))
                tok = int(logits.argmax(dim=-1))
                if tok == 0:
                    break
                toks.append(tok)
                po = pred_of(tok)
        decoded.append(toks)
    return decoded, min_gap


@torch.no_grad()
def greedy_trace(sd: SD, config: dict, enc: Tensor, max_symbols: int = 10):
    """The decisions of `greedy_decode` one by one: a list of (utterance, frame, token the predictor was fed,
    logits (V,)) in the order train.py:180-211 takes them.  Lets a test replay the same decisions on another
    encoder output and compare the gap of exactly the decision that differs."""
    mc = config.get("model", {})
    nl, ks = mc.get("predictor_layers", 2), mc.get("predictor_kernel", 2)
    trace = []
    for b in range(enc.shape[0]):
        last = 0
        for t in range(enc.shape[1]):
            e = enc[b:b + 1, t:t + 1].float()
            for _ in range(max_symbols):
                logits = joint_step(sd, e, predictor_forward(sd, torch.tensor([[last]]), nl, ks))[0]
                trace.append((b, t, last, logits))
                tok = int(logits.argmax())
                if tok == 0:
                    break
                last = tok
    return trace


@torch.no_grad()
def joint_logits(sd: SD, config: dict, enc_t: Tensor, last: int) -> Tensor:
    """Logits (V,) of one decision: encoder frame enc_t (1,1,d), predictor fed `last` (train.py:196-199)."""
    mc = config.get("model", {})
    po = predictor_forward(sd, torch.tensor([[last]]), mc.get("predictor_layers", 2), mc.get("predictor_kernel", 2))
    return joint_step(sd, enc_t.float(), po)[0]


def greedy_decode_tables(enc_proj: Tensor, pred_proj: Tensor, out_w: Tensor, out_b: Tensor, blank: int = 0,
                         max_symbols: int = 10, clamp: float = 15.0) -> Tuple[List[List[int]], float]:
    """The same loop as `greedy_decode` (train.py:180-211) with the joint network of model.py:999-1006 given
    as tables: enc_proj (B,T,dj) = encoder_proj of every frame, pred_proj (V,dj) = predictor_proj of the
    predictor's output for each single token (the predictor of model.py:939-961 is stateless and the loop
    feeds it one token).  Returns the token lists and the smallest top-1/top-2 logit gap."""
    decoded, min_gap = [], float("inf")
    for b in range(enc_proj.shape[0]):
        toks: List[int] = []
        last = 0                                   # train.py:191 starts from token 0
        for t in range(enc_proj.shape[1]):
            for _ in range(max_symbols):
                logits = F.linear(torch.tanh(enc_proj[b, t] + pred_proj[last]), out_w, out_b)
                if clamp > 0:
                    logits = logits.clamp(-clamp, clamp)
                top = logits.topk(2).values
                min_gap = min(min_gap, float(top[0] - top[1]))
                tok = int(logits.argmax())
                if tok == blank:
                    break
                toks.append(tok)
                last = tok
        decoded.append(toks)
    return decoded, min_gap
