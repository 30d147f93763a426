"""Weight pre-packing for the HIP encoder: turns the reference-layout parameters
(`state_dict` names of model.py:779-833) into the layouts the kernels consume.

  * BatchNorm (eval) folded into the preceding convolution:  w' = w * g / sqrt(var + eps),
    b' = (b - mean) * g / sqrt(var + eps) + beta            (model.py:705, :485, eps = 1e-5)
  * 3x3 convs as [Cout][ (kh*3+kw)*Cin + c ] for the channels-last implicit GEMM
  * subsampling Linear columns permuted from c*F' + f (model.py:746-748) to f*C + c
  * q/k/v row-concatenated into one [3*H*64][d] matrix, every head padded to 64 dims with zero
    rows (head_dim 48 of the medium config), out_proj columns padded to match
  * pointwise1 rows interleaved in groups of 16 (a | gate) so the GLU pair lands in one lane
  * depthwise taps as [9][d] fp32
  * the projected relative-position table pos_proj(pos_emb) per frame count T' (weights are
    frozen at inference, so it is computed once per T' with the library's own GEMM and cached)
Packed tensors are plain attributes of a `PackedEncoder`, never part of any state_dict.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List

import torch

from . import _native as N

BN_EPS = 1e-5
HEAD_PAD = 64


def _fold_bn(bn: torch.nn.Module):
    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * s
    return s, shift


def _pad_heads_rows(w: torch.Tensor, heads: int, dh: int) -> torch.Tensor:
    """[H*dh][...] -> [H*64][...] with zero rows after each head's dh rows."""
    rest = w.shape[1:]
    out = w.new_zeros((heads, HEAD_PAD) + tuple(rest))
    out[:, :dh] = w.reshape((heads, dh) + tuple(rest))
    return out.reshape((heads * HEAD_PAD,) + tuple(rest))


class PackedEncoder:
    """Device-resident packed weights of one FastConformerEncoder for one compute dtype."""

    POS_CACHE_ENTRIES = 8         # projected position tables kept (least recently used frame counts are dropped)

    def __init__(self, encoder: torch.nn.Module, dtype: torch.dtype, device: torch.device):
        N.lib()                                   # fail early if the library is not built
        self.dtype, self.device = dtype, device
        self.code = N.dtype_code(dtype)
        sub = encoder.subsampling
        self.d = encoder.d_model
        self.heads = encoder.layers[0].attention.num_heads
        self.dh = self.d // self.heads
        self.hd = self.heads * HEAD_PAD
        self.channels = sub.conv1[0].out_channels
        self.n_mels = sub.d_input
        self.ff = encoder.layers[0].ff1.linear1.out_features
        att0 = encoder.layers[0].attention
        self.window = att0.local_context if att0.use_local_attn else 0
        self.local_context = att0.local_context
        if encoder.layers[0].conv.kernel_size != 9:
            raise NotImplementedError("the HIP depthwise kernel is built for encoder_conv_kernel = 9")
        if self.dh > HEAD_PAD or self.dh % 8 or self.heads % 2:
            raise NotImplementedError(f"head_dim {self.dh} / heads {self.heads}: need head_dim <= 64, % 8 == 0, even heads")
        for v, nm in ((self.d, "d_encoder"), (self.ff, "ff dim"), (self.channels, "subsampling_channels")):
            if v % 128:
                raise NotImplementedError(f"{nm} = {v} must be a multiple of 128 for the MFMA tiles")
        self._keep: List[torch.Tensor] = []       # owns every device tensor the descriptors point to

        def T(t):    # compute-dtype matrix on device
            t = t.detach().to(device=device, dtype=torch.float32).contiguous().to(dtype).contiguous()
            self._keep.append(t)
            return t

        def V(t):    # fp32 vector on device
            t = t.detach().to(device=device, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t

        c = self.channels
        s1, sh1 = _fold_bn(sub.conv1[1])
        w1 = sub.conv1[0].weight.detach().float() * s1.view(-1, 1, 1, 1)          # [C][1][3][3]
        self.conv1_w9 = V(w1.reshape(c, 9).t())                                    # [9][C]
        self.conv1_b = V(sub.conv1[0].bias.detach().float() * s1 + sh1)
        self.conv_w, self.conv_b = [], []
        for blk in (sub.conv2, sub.conv3):
            s, sh = _fold_bn(blk[1])
            w = blk[0].weight.detach().float() * s.view(-1, 1, 1, 1)              # [Co][Ci][3][3]
            self.conv_w.append(T(w.permute(0, 2, 3, 1).reshape(c, 9 * c)))
            self.conv_b.append(V(blk[0].bias.detach().float() * s + sh))
        # conv1 produced inside conv2 (one kernel, no conv1 activation in memory): both weight sets in fragment order
        self.conv12 = None
        self.fuse_conv12 = True          # False: the two-kernel path (A/B timing)
        if N.lib().fcenc_conv12_supported(self.code, self.n_mels, c):
            a1 = torch.empty(N.lib().fcenc_conv12_packed_bytes(0), dtype=torch.uint8, device=device)
            w2 = torch.empty(N.lib().fcenc_conv12_packed_bytes(1), dtype=torch.uint8, device=device)
            N.check(N.lib().fcenc_conv12_pack(N.ptr(self.conv1_w9), N.ptr(self.conv1_b), N.ptr(self.conv_w[0]), N.ptr(a1),
                                              N.ptr(w2), self.n_mels, c, N.stream_of(a1)))
            self._keep += [a1, w2]
            self.conv12 = (a1, w2)
        # conv3 with its input patch staged once in LDS: the weights in fragment order
        self.conv3_patch = None
        self.use_conv3_patch = True      # False: the implicit GEMM (A/B timing)
        f2 = (((self.n_mels + 1) // 2) + 1) // 2
        if N.lib().fcenc_conv3_patch_supported(self.code, f2, c, c):
            wp = torch.empty(N.lib().fcenc_conv3_patch_packed_bytes(), dtype=torch.uint8, device=device)
            N.check(N.lib().fcenc_conv3_patch_pack(N.ptr(self.conv_w[1]), N.ptr(wp), c, N.stream_of(wp)))
            self._keep.append(wp)
            self.conv3_patch = wp
        f3 = sub.linear.in_features // c
        self.f3 = f3
        lw = sub.linear.weight.detach().float().view(self.d, c, f3).permute(0, 2, 1).reshape(self.d, f3 * c)
        self.lin_w, self.lin_b = T(lw), V(sub.linear.bias)

        self.layers: List[Dict[str, torch.Tensor]] = []
        self.pos_w: List[torch.Tensor] = []
        for blk in encoder.layers:
            if blk.conv.norm_type != "batch":
                raise NotImplementedError("the HIP conv module folds BatchNorm; conv_norm_type must be 'batch'")
            a, cv = blk.attention, blk.conv
            L: Dict[str, torch.Tensor] = {}
            for nm, ff in (("ff1", blk.ff1), ("ff2", blk.ff2)):
                L[nm + "_ln_g"], L[nm + "_ln_b"] = V(ff.layer_norm.weight), V(ff.layer_norm.bias)
                w1, w2 = T(ff.linear1.weight), T(ff.linear2.weight)
                if N.lib().fcenc_ffn_fused_supported(self.code, self.d, self.ff):
                    # the fused feed-forward kernel streams its weights in MFMA fragment order
                    w1p, w2p = torch.empty_like(w1), torch.empty_like(w2)
                    N.check(N.lib().fcenc_ffn_pack(N.ptr(w1), N.ptr(w2), N.ptr(w1p), N.ptr(w2p), self.d, self.ff,
                                                   N.stream_of(w1)))
                    self._keep += [w1p, w2p]
                    w1, w2 = w1p, w2p
                L[nm + "_w1"], L[nm + "_b1"] = w1, V(ff.linear1.bias)
                L[nm + "_w2"], L[nm + "_b2"] = w2, V(ff.linear2.bias)
            qkv_w = torch.cat([_pad_heads_rows(p.weight.detach().float(), self.heads, self.dh)
                               for p in (a.q_proj, a.k_proj, a.v_proj)], 0)
            qkv_b = torch.cat([_pad_heads_rows(p.bias.detach().float(), self.heads, self.dh)
                               for p in (a.q_proj, a.k_proj, a.v_proj)], 0)
            L["qkv_w"], L["qkv_b"] = T(qkv_w), V(qkv_b)
            L["qkv_w_rm"] = L["qkv_w"]            # row-major copy for schedules that project with a plain GEMM
            if N.lib().fcenc_ffn_qkv_fused_supported(self.code, self.d, self.ff, self.heads):
                # the first feed-forward kernel also projects q | k | v and streams this matrix like its own
                wq = torch.empty_like(L["qkv_w"])
                N.check(N.lib().fcenc_qkv_pack(N.ptr(L["qkv_w"]), N.ptr(wq), wq.shape[0], wq.shape[1], N.stream_of(wq)))
                self._keep.append(wq)
                L["qkv_w"] = wq
            L["bias_u"] = V(_pad_heads_rows(a.pos_bias_u.detach().float().reshape(-1), self.heads, self.dh))
            L["bias_v"] = V(_pad_heads_rows(a.pos_bias_v.detach().float().reshape(-1), self.heads, self.dh))
            ow = a.out_proj.weight.detach().float()                                # [d][H*dh]
            L["out_w"] = T(_pad_heads_rows(ow.t().contiguous(), self.heads, self.dh).t())
            L["out_w_rm"] = L["out_w"]            # the streaming step multiplies row-major matrices
            if N.lib().fcenc_convmod_oproj_supported(self.code, self.d, 9, self.heads):
                # the fused conv module also applies out_proj and streams this matrix like its own
                wo = torch.empty_like(L["out_w"])
                N.check(N.lib().fcenc_rows_pack(N.ptr(L["out_w"]), N.ptr(wo), wo.shape[0], wo.shape[1], N.stream_of(wo)))
                self._keep.append(wo)
                L["out_w"] = wo
            L["out_b"] = V(a.out_proj.bias)
            self.pos_w.append(T(_pad_heads_rows(a.pos_proj.weight.detach().float(), self.heads, self.dh)))
            L["conv_ln_g"], L["conv_ln_b"] = V(cv.layer_norm.weight), V(cv.layer_norm.bias)
            d = self.d
            pw = cv.pointwise1.weight.detach().float().view(2 * d, d)
            pb = cv.pointwise1.bias.detach().float()
            idx = torch.cat([torch.cat([torch.arange(16 * q, 16 * q + 16), d + torch.arange(16 * q, 16 * q + 16)])
                             for q in range(d // 16)])
            L["pw1_w"], L["pw1_b"] = T(pw[idx]), V(pb[idx])
            L["pw2_w"] = T(cv.pointwise2.weight.detach().float().view(d, d))
            L["pw1_w_rm"], L["pw2_w_rm"] = L["pw1_w"], L["pw2_w"]
            if N.lib().fcenc_convmod_fused_supported(self.code, d, 9):
                # the fused conv module streams its pointwise weights in MFMA fragment order
                w1p, w2p = torch.empty_like(L["pw1_w"]), torch.empty_like(L["pw2_w"])
                N.check(N.lib().fcenc_convmod_pack(N.ptr(L["pw1_w"]), N.ptr(L["pw2_w"]), N.ptr(w1p), N.ptr(w2p), d,
                                                   N.stream_of(w1p)))
                self._keep += [w1p, w2p]
                L["pw1_w"], L["pw2_w"] = w1p, w2p
            s, sh = _fold_bn(cv.norm)
            dw = cv.depthwise.conv.weight.detach().float().view(d, 9) * s.view(-1, 1)
            L["dw_w9"] = V(dw.t())
            L["dw_b"] = V(cv.depthwise.conv.bias.detach().float() * s + sh)
            L["pw2_b"] = V(cv.pointwise2.bias)
            L["ln_g"], L["ln_b"] = V(blk.layer_norm.weight), V(blk.layer_norm.bias)
            self.layers.append(L)
        self._pe = encoder.pos_encoding
        # optional HIP events (begin, end[, begin, end]) recorded around the conv2 launch and around the launch that
        # follows the attention core of the middle block
        self.profile_events = None
        self.workspace: dict = {}                 # scratch of the offline pass, grown on demand, reused across calls
        self.stream_workspace: dict = {}          # scratch of eager streaming steps (captured steps: StreamCache.workspace)
        from . import ops
        self.handle = ops.register_handle(self)   # names this object in the fcenc::encoder_* operators
        self._pos_cache: Dict[int, List[torch.Tensor]] = {}
        self._desc_cache: Dict[int, tuple] = {}
        self._stream_pos_cache: Dict[int, List[torch.Tensor]] = {}
        self._lane_cache: Dict[int, tuple] = {}

    # ------------------------------------------------------------------------------------
    def pos_tables(self, t_valid: int) -> List[torch.Tensor]:
        """Per layer: [2*Tp+64][H*64] table with pos_proj(pos_emb)[r] at row r+64, r = (T'-1)+(j-i)."""
        if t_valid in self._pos_cache:
            self._pos_cache[t_valid] = self._pos_cache.pop(t_valid)         # most recently used last
        else:
            from . import ops
            # a table set is n_layers x [2*Tp+64][H*64]: a few MB each; batches padded to many different lengths
            # (train.py:174 pads to the batch maximum) would otherwise pile up one set per length
            while len(self._pos_cache) >= self.POS_CACHE_ENTRIES:
                old = next(iter(self._pos_cache))
                del self._pos_cache[old]
                self._desc_cache.pop(old, None)
            tp = (t_valid + 63) // 64 * 64
            pos_emb = self._pe.table(t_valid).to(device=self.device, dtype=torch.float32)    # [2T'-1][d]
            ext = torch.zeros((2 * tp + 64, self.d), dtype=torch.float32, device=self.device)
            ext[64:64 + 2 * t_valid - 1] = pos_emb
            ext = ext.to(self.dtype).contiguous()
            self._pos_cache[t_valid] = [ops.linear(ext, w, None, N.EPI_STORE) for w in self.pos_w]
        return self._pos_cache[t_valid]

    def descriptor(self, t_valid: int):
        """(Encoder struct, keep-alive tuple) for frame count T' (0: weights only, no position tables)."""
        pos = self.pos_tables(t_valid) if t_valid > 0 else None       # also marks this frame count as recently used
        if t_valid not in self._desc_cache:
            arr = (N.Layer * len(self.layers))()
            for i, L in enumerate(self.layers):
                for name in N.Layer._names:
                    if name == "pos":
                        arr[i].pos = pos[i].data_ptr() if pos is not None else None
                    else:
                        setattr(arr[i], name, L[name].data_ptr())
            e = N.Encoder()
            e.dtype, e.d_model, e.n_heads, e.head_dim = self.code, self.d, self.heads, self.dh
            e.n_layers, e.ff_dim, e.channels, e.n_mels = len(self.layers), self.ff, self.channels, self.n_mels
            e.window = self.window
            e.conv1_w9, e.conv1_b = self.conv1_w9.data_ptr(), self.conv1_b.data_ptr()
            e.conv2_w, e.conv2_b = self.conv_w[0].data_ptr(), self.conv_b[0].data_ptr()
            e.conv3_w, e.conv3_b = self.conv_w[1].data_ptr(), self.conv_b[1].data_ptr()
            e.lin_w, e.lin_b = self.lin_w.data_ptr(), self.lin_b.data_ptr()
            e.layers = C.cast(arr, C.POINTER(N.Layer))
            e.ev_begin = e.ev_end = e.ev_block_begin = e.ev_block_end = None
            e.conv12_a1 = e.conv12_w2 = e.conv3_wp = None
            self._desc_cache[t_valid] = (e, arr)
        return self._desc_cache[t_valid]

    # ------------------------------------------------------------------------------------
    def forward(self, mel: torch.Tensor, lengths, workspace_cache: dict, events=None, lanes: int = 0):
        """mel [B][T][n_mels] fp32 on this device -> (out fp32 [B][T'][d], out_len int64 [B], mask bool or None).
        lanes: slices of the batch whose conformer blocks run on concurrent HIP streams (0 = by batch size)."""
        N.require(mel, torch.float32, "mel", 3)
        if mel.device != self.device:
            raise RuntimeError(f"fcenc: weights are packed on {self.device}, input is on {mel.device}")
        b, t, f = mel.shape
        if f != self.n_mels:
            raise ValueError(f"fcenc: expected {self.n_mels} mel bins, got {f}")
        if lengths is not None:
            lengths = lengths.to(device=self.device, dtype=torch.int64).contiguous()
            if lengths.numel() != b:
                raise ValueError("fcenc: lengths must have one entry per utterance")
        lib = N.lib()
        t_out = lib.fcenc_out_frames(t)
        enc, _keep = self.descriptor(t_out)
        fused = self.conv12 if self.fuse_conv12 else None
        enc.conv12_a1, enc.conv12_w2 = (fused[0].data_ptr(), fused[1].data_ptr()) if fused else (None, None)
        enc.conv3_wp = self.conv3_patch.data_ptr() if self.conv3_patch is not None and self.use_conv3_patch else None
        need = lib.fcenc_encoder_workspace_bytes(C.byref(enc), b, t)
        if need == 0:
            raise RuntimeError("fcenc: " + lib.fcenc_last_error().decode())
        ws = workspace_cache.get("ws")
        if ws is None or ws.numel() < need or ws.device != self.device:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            workspace_cache["ws"] = ws
        out = torch.empty((b, t_out, self.d), dtype=torch.float32, device=self.device)
        out_len = torch.empty((b,), dtype=torch.int64, device=self.device)
        mask = torch.empty((b, t_out), dtype=torch.uint8, device=self.device) if lengths is not None else None
        events = events if events is not None else self.profile_events
        events = tuple(events) + (None,) * 4 if events is not None else (None,) * 4
        enc.ev_begin, enc.ev_end, enc.ev_block_begin, enc.ev_block_end = events[:4]
        n_lanes = max(1, min(lanes or self.default_forward_lanes(b), b, 8))
        side = C.byref(self._side_lanes(n_lanes - 1)[0]) if n_lanes > 1 else None
        N.check(lib.fcenc_encoder_forward_lanes(C.byref(enc), mel.data_ptr(), N.ptr(lengths), b, t, out.data_ptr(),
                                                out_len.data_ptr(), N.ptr(mask), ws.data_ptr(), ws.numel(),
                                                N.stream_of(mel), side))
        enc.ev_begin = enc.ev_end = enc.ev_block_begin = enc.ev_block_end = None
        return out, out_len, (mask.view(torch.bool) if mask is not None else None)

    # ------------------------------------------------------------------------------------
    # cached streaming step
    def stream_supported(self, t_out: int) -> bool:
        lib = N.lib()
        return (1 <= t_out <= lib.fcenc_stream_max_chunk_frames() and self.local_context % 8 == 0
                and t_out <= self.local_context <= lib.fcenc_stream_max_cache_frames())

    def stream_pos_tables(self, t_out: int) -> List[torch.Tensor]:
        """Per layer: pos_proj(pos_emb) [2*Tq-1][H*64] for a chunk of Tq frames (model.py:175-180, :376)."""
        if t_out not in self._stream_pos_cache:              # at most stream_max_chunk_frames (16) entries of a few KB
            from . import ops
            pos_emb = self._pe.table(t_out).to(device=self.device, dtype=torch.float32).to(self.dtype).contiguous()
            self._stream_pos_cache[t_out] = [ops.linear(pos_emb, w, None, N.EPI_STORE) for w in self.pos_w]
        return self._stream_pos_cache[t_out]

    def new_stream_cache(self, batch: int):
        from .streaming import StreamCache
        return StreamCache(self, batch)

    def _side_lanes(self, n_side: int):
        """(StreamLanes struct, keep-alive) with n_side side streams and their fork / join events."""
        if n_side not in self._lane_cache:
            lanes = N.StreamLanes()
            lanes.n = n_side
            streams = [torch.cuda.Stream(device=self.device) for _ in range(n_side)]
            for i, st in enumerate(streams):
                lanes.streams[i] = st.cuda_stream
            for i in range(n_side + 1):
                ev = N.lib().fcenc_event_create()
                if not ev:
                    raise RuntimeError("fcenc: " + N.lib().fcenc_last_error().decode())
                lanes.events[i] = ev
            self._lane_cache[n_side] = (lanes, streams)
        return self._lane_cache[n_side]

    @staticmethod
    def default_forward_lanes(batch: int) -> int:
        """Lanes the offline forward uses when the caller does not say."""
        return 1

    @staticmethod
    def default_stream_lanes(batch: int) -> int:
        """Lanes the streaming step uses when the caller does not say.  Three for 768 streams or more: a 1024-stream
        chunk measured 2.59 ms against 2.75 ms with one lane (2.78 against 2.92 ms with the mel and decode stages
        around it), 768 streams 2.37 against 2.43 ms, 1536 streams 3.31 against 3.56 ms -- one lane's cache-bound
        attention overlaps another's weight-bound matrix products.  At 512 streams lanes change nothing (2.03 ms),
        below they cost (256 streams: 1.71 -> 1.84 ms), so smaller batches stay on one lane.  Four lanes give 2.67 ms
        at 1024 streams (the host queues that many times the launches); six and more exceed the four hardware queues
        of a process and are slower than one."""
        return 3 if batch >= 768 else 1

    def stream_step(self, mel: torch.Tensor, state, workspace_cache: dict, lanes: int = 0,
                    graph: bool = False) -> torch.Tensor:
        """One chunk for every stream: mel [B][T][n_mels] fp32 -> out fp32 [B][T'][d]; `state` (StreamCache)
        advances.  lanes: slices of the batch run concurrently on side HIP streams (0 = by batch size).
        graph: from the second chunk of a given length on, the step's launches (a few hundred) are captured once in
        a HIP graph and replayed; the ring position then lives in device memory (`state.ring`) and the step moves
        it on itself."""
        N.require(mel, torch.float32, "mel", 3)
        if mel.device != self.device:
            raise RuntimeError(f"fcenc: weights are packed on {self.device}, input is on {mel.device}")
        b, t, f = mel.shape
        if f != self.n_mels:
            raise ValueError(f"fcenc: expected {self.n_mels} mel bins, got {f}")
        if state.packed is not self or b != state.batch:
            raise ValueError("fcenc: the streaming cache belongs to another encoder / batch size")
        lib = N.lib()
        t_out = lib.fcenc_out_frames(t)
        n_lanes = max(1, min(lanes or self.default_stream_lanes(b), b, 8))
        key = (t, n_lanes)
        if graph:
            # a captured step has the workspace's addresses baked in: a replaying state owns its workspace, so that
            # no other session of this encoder (another batch size, chunk length or lane count) can reallocate it
            workspace_cache = state.workspace
        if graph and key in state.graphs:
            g, g_mel, g_out, _g_ws = state.graphs[key]
            if state.ring_chunk != t_out:
                state.sync_ring(t_out)
            g_mel.copy_(mel)
            g.replay()
            state.c.frames += t_out                  # a replay runs the device work only
            return g_out.clone()
        enc, _keep = self.descriptor(0)           # weights only; the position table comes from the state
        state.set_pos(self.stream_pos_tables(t_out))
        need = lib.fcenc_encoder_stream_workspace_bytes(C.byref(enc), b, t, n_lanes)
        if need == 0:
            raise RuntimeError("fcenc: " + lib.fcenc_last_error().decode())
        ws = workspace_cache.get("ws")
        if ws is None or ws.numel() < need or ws.device != self.device:
            if state.graphs:
                state.graphs.clear()              # captured steps point into the old workspace
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            workspace_cache["ws"] = ws
        side = C.byref(self._side_lanes(n_lanes - 1)[0]) if n_lanes > 1 else None

        def launch(src, dst):
            N.check(lib.fcenc_encoder_stream_step(C.byref(enc), C.byref(state.c), src.data_ptr(), t, dst.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), N.stream_of(src), side))

        if graph and state.frames > 0:
            # capture (nothing runs), then replay once: together one step.  The first chunk of a stream always runs
            # eagerly, so every kernel has been loaded and configured before a capture begins.
            state.sync_ring(t_out)
            state.c.ring = state.ring.data_ptr()
            g_mel = mel.clone()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    g_out = torch.empty((b, t_out, self.d), dtype=torch.float32, device=self.device)
                    launch(g_mel, g_out)
            finally:
                state.c.ring = None
            state.graphs[key] = (g, g_mel, g_out, ws)    # the graph keeps its workspace alive
            g.replay()
            return g_out.clone()
        state.ring_chunk = 0                          # an eager step moves the host count only
        out = torch.empty((b, t_out, self.d), dtype=torch.float32, device=self.device)
        launch(mel, out)
        return out
