"""Device-resident state of the cached streaming encoder (FastConformerEncoder.forward with
`cache=` / `streaming=True`, model.py:835-894; caller inference.py:358).

The reference carries, per layer, the attention keys | values of the last `local_context` frames as
one (B, kv, 2d) tensor that is re-concatenated and re-sliced at every chunk (model.py:352-366) and the
last 8 GLU frames (B, d, 8) of the depthwise conv (model.py:236-249).  Here the same information lives
in ring buffers that the kernels update in place: K as [B][H][cap][64] and V as [B][H][cap/8][64][8]
(blocks of 8 slots, one 8-slot group per value dimension) in the compute dtype -- the layouts the
attention kernel reads straight into MFMA fragments -- with frame t of a stream in slot t % cap, and
the conv frames as [B][8][d].  All streams of a batch advance in lock step.

A `StreamCache` is what the encoder returns as `new_cache` on the native path.  It is passed back
unchanged with the next chunk; code that wants the reference's tensors can index or iterate it (it
then behaves like the reference's list of (attn_cache, conv_cache) tuples) or call `to_reference()`.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch

from . import _native as N


def v_to_slots(v: torch.Tensor) -> torch.Tensor:
    """V storage [B][H][cap/8][64][8] -> a [B][H][cap][64] (slot, dim) copy."""
    b, h, nblk = v.shape[:3]
    return v.permute(0, 1, 2, 4, 3).reshape(b, h, nblk * 8, 64)


def v_from_slots(x: torch.Tensor) -> torch.Tensor:
    """[B][H][cap][64] (slot, dim) -> V storage order [B][H][cap/8][64][8] (contiguous copy)."""
    b, h, cap = x.shape[:3]
    return x.reshape(b, h, cap // 8, 8, 64).permute(0, 1, 2, 4, 3).contiguous()


class StreamCache:
    def __init__(self, packed, batch: int):
        self.packed = packed
        self.batch = batch
        self.cap = packed.local_context
        dev, dt = packed.device, packed.dtype
        n = len(packed.layers)
        h = packed.heads
        self.k = [torch.zeros((batch, h, self.cap, 64), dtype=dt, device=dev) for _ in range(n)]
        self.v = [torch.zeros((batch, h, self.cap // 8, 64, 8), dtype=dt, device=dev) for _ in range(n)]
        self.conv = [torch.zeros((batch, 8, packed.d), dtype=dt, device=dev) for _ in range(n)]
        self._arr = (N.StreamLayer * n)()
        for i, L in enumerate(packed.layers):
            a = self._arr[i]
            a.kcache, a.vcache, a.conv_cache = self.k[i].data_ptr(), self.v[i].data_ptr(), self.conv[i].data_ptr()
            a.out_w, a.pw1_w, a.pw2_w = L["out_w_rm"].data_ptr(), L["pw1_w_rm"].data_ptr(), L["pw2_w_rm"].data_ptr()
            a.qkv_w = L["qkv_w_rm"].data_ptr()
            a.pos = None
        self.c = N.StreamState()
        self.c.batch, self.c.cap, self.c.frames = batch, self.cap, 0
        self.c.layers = C.cast(self._arr, C.POINTER(N.StreamLayer))
        self.c.ring = None
        # HIP-graph replay of the step (PackedEncoder.stream_step(graph=True)): the ring position lives on the device
        self.ring: Optional[torch.Tensor] = None      # int32 {frames % cap, min(frames + T', cap), min(frames, cap)}
        self.ring_chunk = 0                           # T' the device copy was derived for; 0: out of date
        self.graphs: dict = {}                        # (T, lanes) -> (graph, static input, static output, workspace)
        self.workspace: dict = {}                     # workspace of the captured steps (never shared between states)
        from . import ops
        self.handle = ops.register_handle(self)       # names this state in fcenc::encoder_stream_step

    def sync_ring(self, t_out: int) -> None:
        """Device copy of the ring position from the host's frame count, for chunks of `t_out` frames."""
        f = int(self.c.frames)
        vals = torch.tensor([f % self.cap, min(f + t_out, self.cap), min(f, self.cap)], dtype=torch.int32)
        if self.ring is None:
            self.ring = torch.zeros(3, dtype=torch.int32, device=self.packed.device)
        self.ring.copy_(vals, non_blocking=False)
        self.ring_chunk = t_out

    @property
    def frames(self) -> int:
        """Frames every stream has seen so far."""
        return int(self.c.frames)

    def set_pos(self, tables: List[torch.Tensor]) -> None:
        for i, t in enumerate(tables):
            self._arr[i].pos = t.data_ptr()

    # ---- exchange with the reference's cache format -------------------------------------------
    @classmethod
    def from_reference(cls, packed, cache, batch: int) -> "StreamCache":
        """cache: list of (attn_cache (B, kv, 2d) | None, conv_cache (B, d, 8) | None) per layer."""
        st = cls(packed, batch)
        if len(cache) != len(packed.layers):
            raise ValueError(f"cache has {len(cache)} layers, the encoder {len(packed.layers)}")
        kv_len = None
        d, h, dh = packed.d, packed.heads, packed.dh
        for i, (ac, cc) in enumerate(cache):
            n = 0 if ac is None else ac.shape[1]
            if kv_len is None:
                kv_len = n
            if n != kv_len:
                raise ValueError("all layers of a streaming cache must hold the same number of frames")
            if n > st.cap:
                raise ValueError(f"attention cache of {n} frames exceeds local_context = {st.cap}")
            if ac is not None:
                if ac.shape[0] != batch or ac.shape[2] != 2 * d:
                    raise ValueError(f"layer {i}: attention cache must be (B, kv, 2*d_model), got {tuple(ac.shape)}")
                kk, vv = ac.to(packed.device).float().chunk(2, dim=-1)
                st.k[i][:, :, :n, :dh] = kk.reshape(batch, n, h, dh).permute(0, 2, 1, 3).to(packed.dtype)
                slots = torch.zeros((batch, h, st.cap, 64), dtype=packed.dtype, device=packed.device)
                slots[:, :, :n, :dh] = vv.reshape(batch, n, h, dh).permute(0, 2, 1, 3).to(packed.dtype)
                st.v[i].copy_(v_from_slots(slots))
            if cc is not None:
                if tuple(cc.shape) != (batch, d, 8):
                    raise ValueError(f"layer {i}: conv cache must be (B, d_model, 8), got {tuple(cc.shape)}")
                st.conv[i].copy_(cc.to(packed.device).float().transpose(1, 2).to(packed.dtype))
        st.c.frames = kv_len or 0
        return st

    def to_reference(self) -> List[Tuple[Optional[torch.Tensor], torch.Tensor]]:
        """The same state as the reference returns it: per layer (cat(k, v) (B, kv, 2d) fp32, conv (B, d, 8) fp32)."""
        p = self.packed
        frames = self.frames
        kv = min(frames, self.cap)
        slots = torch.arange(frames - kv, frames, device=p.device) % self.cap
        out = []
        for k, v, cc in zip(self.k, self.v, self.conv):
            if kv == 0:
                ac = None
            else:
                kk = k[:, :, slots, :p.dh].permute(0, 2, 1, 3).reshape(self.batch, kv, p.d)
                vv = v_to_slots(v)[:, :, slots, :p.dh].permute(0, 2, 1, 3).reshape(self.batch, kv, p.d)
                ac = torch.cat([kk, vv], dim=-1).float()
            out.append((ac, cc.float().transpose(1, 2).contiguous()))
        return out

    def __len__(self) -> int:
        return len(self.k)

    def __getitem__(self, i):
        return self.to_reference()[i]

    def __iter__(self):
        return iter(self.to_reference())
