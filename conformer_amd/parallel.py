"""Utterance sharding across the GPUs of one node.

In eval mode every encoder op is per utterance (BatchNorm uses running statistics, the input
standardisation is per utterance, pos_proj(pos_emb) is batch independent), so a batch shards on
dim 0 with a full weight replica per GPU and needs ONE exchange: an all-gather of the encoder
outputs and lengths.  The batch must be padded to its global maximum BEFORE it is cut: the
reference's per-utterance mean/std and the unmasked stride-2 convs see the zero padding
(model.py:857), so re-padding a shard would change its numbers.

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm (the reference's own
init pattern, utils.py:245-258), "gloo" on CPU for tests.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from RANK / WORLD_SIZE / LOCAL_RANK; initialises the default
    process group when WORLD_SIZE > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            # FCENC_DIST_BACKEND=gloo: rehearsal of the multi-rank path on fewer GPUs than ranks (ranks then share
            # devices, which RCCL refuses; the gather is staged through host memory, see gather_outputs_async)
            backend = os.environ.get("FCENC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of the batch for `rank`; the first batch % world ranks take one extra."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(mel: Tensor, lengths: Optional[Tensor], world: int, rank: int):
    """Cut the ALREADY PADDED (B, T, F) batch and its lengths for this rank (views, no re-padding)."""
    lo, hi = shard_bounds(mel.size(0), world, rank)
    return mel[lo:hi], (None if lengths is None else lengths[lo:hi])


class PendingGather:
    """An all-gather of outputs and lengths in flight (`gather_outputs_async`).  `wait()` makes the current
    stream wait for it (the host does not block on RCCL) and returns ((B, T', d), (B,))."""

    def __init__(self, all_out: Tensor, all_len: Tensor, works, sizes, big: int):
        self._out, self._len, self._works, self._sizes, self._big = all_out, all_len, works, sizes, big

    def wait(self) -> Tuple[Tensor, Tensor]:
        for w in self._works:
            w.wait()
        self._works = []
        all_out, all_len = self._out, self._len
        if min(self._sizes) != self._big:
            keep = torch.cat([torch.arange(r * self._big, r * self._big + n) for r, n in enumerate(self._sizes)])
            keep = keep.to(all_out.device)
            all_out, all_len = all_out.index_select(0, keep), all_len.index_select(0, keep)
        return all_out, all_len


def gather_outputs_async(out: Tensor, out_len: Tensor, batch: int, group=None) -> PendingGather:
    """Starts the all-gather of per-rank encoder outputs (b_r, T', d) and lengths (b_r,) in rank order and
    returns at once; the collective runs on the backend's own stream behind the work already queued on the
    current one, so the caller can queue the next forward beside it and `wait()` when it needs the result.
    Equal shards use one all_gather_into_tensor per tensor; ragged shards are padded to the largest shard for
    the exchange and trimmed in `wait()`."""
    world = dist.get_world_size(group)
    sizes = [shard_bounds(batch, world, r)[1] - shard_bounds(batch, world, r)[0] for r in range(world)]
    big = max(sizes)
    if out.size(0) != big:
        pad = big - out.size(0)
        out = torch.cat([out, out.new_zeros((pad,) + tuple(out.shape[1:]))], 0)
        out_len = torch.cat([out_len, out_len.new_zeros((pad,))], 0)
    if out.is_cuda and dist.get_backend(group) == "gloo":
        # gloo moves host memory only: stage through it (rehearsals with ranks sharing a GPU; never the RCCL path)
        dev = out.device
        h_out, h_len = out.contiguous().cpu(), out_len.contiguous().cpu()
        g_out, g_len = h_out.new_empty((world * big,) + tuple(out.shape[1:])), h_len.new_empty((world * big,))
        dist.all_gather_into_tensor(g_out, h_out, group=group)
        dist.all_gather_into_tensor(g_len, h_len, group=group)
        return PendingGather(g_out.to(dev), g_len.to(dev), [], sizes, big)
    all_out = out.new_empty((world * big,) + tuple(out.shape[1:]))
    all_len = out_len.new_empty((world * big,))
    works = [dist.all_gather_into_tensor(all_out, out.contiguous(), group=group, async_op=True),
             dist.all_gather_into_tensor(all_len, out_len.contiguous(), group=group, async_op=True)]
    return PendingGather(all_out, all_len, works, sizes, big)


def gather_outputs(out: Tensor, out_len: Tensor, batch: int, group=None) -> Tuple[Tensor, Tensor]:
    """All-gather of per-rank encoder outputs (b_r, T', d) and lengths (b_r,) in rank order ->
    ((B, T', d), (B,)) on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return out, out_len
    return gather_outputs_async(out, out_len, batch, group).wait()


def encode_sharded(model, mel: Tensor, lengths: Optional[Tensor], group=None) -> Tuple[Tensor, Tensor]:
    """Every rank passes the same globally padded batch (or at least its own slice of it);
    returns the full (B, T', d) outputs and (B,) lengths on every rank."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    m, ln = shard_batch(mel, lengths, world, rank)
    if m.size(0) == 0:      # more ranks than utterances: this rank only takes part in the gather
        t_out = mel.size(1)
        for _ in range(3):
            t_out = (t_out + 1) // 2
        out = mel.new_zeros((0, t_out, model.d_encoder), dtype=torch.float32)
        out_len = torch.zeros((0,), dtype=torch.int64, device=mel.device)
    else:
        out, out_len, _ = model.encode(m, ln)
    return gather_outputs(out, out_len, mel.size(0), group)
