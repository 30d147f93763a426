"""Greedy RNN-T decode on the device: drop-in for `greedy_decode(model, encoder_out, max_symbols)` of
train.py:180-211 and `_greedy_decode` of inference.py:161-194.

The reference walks utterance by utterance, frame by frame, symbol by symbol in Python, with one
`.item()` synchronisation and a predictor + joint call per decision.  Its predictor is stateless and
is fed only the last emitted token, so predictor_proj(predictor([v])) is a table over the vocabulary
(computed here with the model's own PyTorch modules, token by token exactly as the reference calls
them), encoder_proj of all frames is one GEMM (`fcenc_gemm`, exact-fp32 MFMA), and the argmax loop of a
whole batch is one kernel launch (`fcenc_greedy_decode`, one workgroup per utterance).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import _native as N
from . import ops


def _joint_tables(model, device: torch.device):
    """(pred_proj [V][dj], enc_w [djp][d], enc_b [djp], out_w [V][dj], out_b [V]) fp32 on `device`, cached on
    the model until one of the parameters involved changes."""
    params = list(model.predictor.parameters()) + list(model.joint.parameters())
    key = (str(device),) + tuple((p.data_ptr(), p._version) for p in params)
    cached = getattr(model, "_fcenc_joint_tables", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    joint = model.joint
    vocab = joint.output_proj.out_features
    dj = joint.encoder_proj.out_features
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            rows = []
            for v in range(vocab):          # one token at a time, as the reference's loop calls the predictor
                y = torch.tensor([[v]], dtype=torch.long, device=next(model.predictor.parameters()).device)
                rows.append(joint.predictor_proj(model.predict(y)).reshape(dj))
            pred_proj = torch.stack(rows).to(device=device, dtype=torch.float32).contiguous()
            djp = (dj + 127) // 128 * 128     # the GEMM's feature tile
            enc_w = torch.zeros((djp, joint.encoder_proj.in_features), dtype=torch.float32, device=device)
            enc_w[:dj] = joint.encoder_proj.weight.detach().to(device=device, dtype=torch.float32)
            enc_b = torch.zeros((djp,), dtype=torch.float32, device=device)
            enc_b[:dj] = joint.encoder_proj.bias.detach().to(device=device, dtype=torch.float32)
            out_w = joint.output_proj.weight.detach().to(device=device, dtype=torch.float32).contiguous()
            out_b = joint.output_proj.bias.detach().to(device=device, dtype=torch.float32).contiguous()
    finally:
        model.train(was_training)
    tables = (pred_proj, enc_w, enc_b, out_w, out_b)
    model._fcenc_joint_tables = (key, tables)
    return tables


def greedy_decode_device(model, encoder_out: torch.Tensor, max_symbols: int = 100,
                         last_tokens: Optional[torch.Tensor] = None):
    """-> (tokens int32 [B][T * max_symbols], counts int32 [B], last int32 [B]) on the device, no synchronisation.
    `last_tokens` (int32 [B], default all 0) is the token each utterance's predictor starts from; the returned
    `last` is the one it ended on, to be passed back with the stream's next chunk of encoder frames."""
    if not encoder_out.is_cuda:
        raise RuntimeError("conformer_amd: greedy_decode runs on MI355X through libfcenc_hip.so only; "
                           "move the encoder output to the GPU (there is no CPU execution path)")
    if encoder_out.dim() != 3:
        raise ValueError(f"encoder_out must be (batch, T, d_encoder), got {tuple(encoder_out.shape)}")
    b, t, d = encoder_out.shape
    pred_proj, enc_w, enc_b, out_w, out_b = _joint_tables(model, encoder_out.device)
    if d != enc_w.shape[1]:
        raise ValueError(f"encoder_out has width {d}, the joint network expects {enc_w.shape[1]}")
    if d % 32:
        raise NotImplementedError("d_encoder must be a multiple of 32 for the fp32 GEMM")
    x = encoder_out.detach().to(torch.float32).contiguous().view(b * t, d)
    enc_proj = ops.linear(x, enc_w, enc_b, N.EPI_STORE).view(b, t, enc_w.shape[0])
    clamp = float(getattr(model.joint, "logit_clamp", 0.0) or 0.0)
    return ops.greedy_decode(enc_proj, pred_proj, out_w, out_b, out_w.shape[1], int(model.blank_id), int(max_symbols),
                             clamp, t * int(max_symbols), last_tokens)


def _to_lists(tokens: torch.Tensor, counts: torch.Tensor) -> List[List[int]]:
    tokens, counts = tokens.cpu(), counts.cpu().tolist()       # the only synchronisation
    return [tokens[i, :n].tolist() for i, n in enumerate(counts)]


def greedy_decode(model, encoder_out: torch.Tensor, max_symbols: int = 100) -> List[List[int]]:
    """Same result as the reference's `greedy_decode(model, encoder_out, max_symbols)`: a list of token lists."""
    tokens, counts, _ = greedy_decode_device(model, encoder_out, max_symbols)
    return _to_lists(tokens, counts)


class IncrementalGreedyDecoder:
    """The decoding half of the reference's streaming loop (inference.py:305-315 reset, :379-406
    `_decode_incremental`) for a batch of concurrent streams: the predictor state a stream carries from chunk to
    chunk is its last emitted token, kept here on the device.

        dec = IncrementalGreedyDecoder(model, n_streams)
        for chunk in chunks:
            enc, _, cache = model.encode(chunk, None, cache=cache, streaming=True)
            new_tokens = dec.step(enc)          # list of n_streams lists
    """

    def __init__(self, model, batch: int, max_symbols_per_step: int = 10):
        self.model, self.batch, self.max_symbols = model, batch, max_symbols_per_step
        self.last: Optional[torch.Tensor] = None
        self.decoded: List[List[int]] = [[] for _ in range(batch)]

    def reset(self) -> None:
        self.last = None
        self.decoded = [[] for _ in range(self.batch)]

    def step_device(self, encoder_out: torch.Tensor):
        """-> (tokens int32 [B][T * max_symbols], counts int32 [B]) of this chunk, no synchronisation."""
        if encoder_out.shape[0] != self.batch:
            raise ValueError(f"decoder holds {self.batch} streams, got {encoder_out.shape[0]}")
        tokens, counts, self.last = greedy_decode_device(self.model, encoder_out, self.max_symbols, self.last)
        return tokens, counts

    def step(self, encoder_out: torch.Tensor) -> List[List[int]]:
        new = _to_lists(*self.step_device(encoder_out))
        for acc, n in zip(self.decoded, new):
            acc.extend(n)
        return new
