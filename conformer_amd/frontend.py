"""Mel-spectrogram front-end on the device: drop-in for the `torchaudio.transforms.MelSpectrogram` objects the
reference builds (train.py:57-61, 67-71, inference.py:88-92: sample_rate, n_mels, hop_length with n_fft = 400;
inference.py:294-300: n_fft = 512, win_length = 400) and applies per utterance before the encoder
(train.py:122, inference.py:134, 349).

    mel = MelSpectrogram(sample_rate=16000, n_mels=80, hop_length=160).cuda()
    m = mel(waveform)                 # (..., samples) -> (..., n_mels, frames), like torchaudio
    x, frames = mel.batch(waves, n)   # padded (B, S) + sample counts -> (B, T, n_mels) + frame counts: what
                                      # dataset + collate hand to the encoder (train.py:122, 170-174)

One kernel launch (`fcenc_mel_spectrogram`): framing, reflect padding, window, DFT on the exact-fp32 matrix
cores, power, filter bank, written in the encoder's (B, T, n_mels) layout.  The other MelSpectrogram arguments
stay at the defaults the reference uses (power 2, centre, reflect, HTK scale, no normalisation).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from . import _native as N

Tensor = torch.Tensor


def mel_filter_bank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> Tensor:
    """HTK triangular filters without normalisation, (n_freqs, n_mels) fp32: the published definition of
    torchaudio.functional.melscale_fbanks, evaluated in fp32 like it."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    f_pts = 700.0 * (10.0 ** (torch.linspace(m_min, m_max, n_mels + 2) / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    return torch.clamp(torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0)


def plan_geometry(n_fft: int) -> Dict[str, int]:
    """Row and column counts of the twice-folded DFT matrix (include/fcenc.h)."""
    q = n_fft // 4
    ch = 8 if n_fft == 400 else 12
    b0, b1, b2 = (q + 1 + 15) // 16, (q + 15) // 16, (q - 1 + 15) // 16
    ncol = 16 * (b0 + 2 * b1 + b2)
    return {"q": q, "kp": (q + 1 + ch - 1) // ch * ch, "b0": b0, "b1": b1, "b2": b2, "b3": b1,
            "wpitch": ncol if ncol % 64 in (16, 48) else ncol + 16}


def build_plan(sample_rate: int, n_fft: int, win_length: int, hop_length: int, n_mels: int, f_min: float,
               f_max: float) -> Dict[str, object]:
    """Host tensors of `fcenc_mel_plan` (layouts in include/fcenc.h)."""
    if n_fft not in (400, 512):
        raise NotImplementedError(f"n_fft {n_fft}: the kernel is built for 400 and 512")
    if win_length > n_fft or (n_fft - win_length) % 2:
        raise ValueError(f"win_length {win_length} must be <= n_fft {n_fft} and differ from it by an even count")
    lo = (n_fft - win_length) // 2
    half = n_fft // 2
    win = torch.zeros(n_fft, dtype=torch.float32)
    win[lo:lo + win_length] = torch.hann_window(win_length, periodic=True, dtype=torch.float32)
    if not torch.allclose(win[1:half], win[half + 1:].flip(0), rtol=0.0, atol=1e-6):    # fp32 Hann: equal to an ulp
        raise ValueError("the window must be symmetric about n_fft / 2")
    geo = plan_geometry(n_fft)
    q, kp, b0, b1, b2 = geo["q"], geo["kp"], geo["b0"], geo["b1"], geo["b2"]
    m = torch.arange(kp, dtype=torch.float64)[:, None]
    live = (m <= q).double()
    c = torch.where(m == q, 0.5, 1.0) * live                     # the sample that is its own mirror counts once
    w = 2 * math.pi / n_fft
    dft = torch.zeros(kp, geo["wpitch"], dtype=torch.float64)
    j = torch.arange(0, q + 1, dtype=torch.float64)[None, :]
    dft[:, :q + 1] = c * torch.cos(w * (2 * j) * m)
    j = torch.arange(0, q, dtype=torch.float64)[None, :]
    o = 16 * b0
    dft[:, o:o + q] = live * torch.cos(w * (2 * j + 1) * m)
    o += 16 * b1
    dft[:, o:o + q - 1] = live * torch.sin(w * (2 * j[:, :q - 1] + 2) * m)
    o += 16 * b2
    dft[:, o:o + q] = c * torch.sin(w * (2 * j + 1) * m)
    fb = mel_filter_bank(half + 1, f_min, f_max, n_mels, sample_rate)
    start = torch.zeros(n_mels, dtype=torch.int32)
    count = torch.zeros(n_mels, dtype=torch.int32)
    for f in range(n_mels):
        nz = torch.nonzero(fb[:, f]).flatten()
        if nz.numel():
            start[f], count[f] = int(nz[0]), int(nz[-1]) - int(nz[0]) + 1
    maxw = max(int(count.max()), 1)
    fbw = torch.zeros(n_mels, maxw, dtype=torch.float32)
    for f in range(n_mels):
        fbw[f, :int(count[f])] = fb[int(start[f]):int(start[f]) + int(count[f]), f]
    return {"n_fft": n_fft, "hop": hop_length, "kp": kp, "wpitch": geo["wpitch"], "n_mels": n_mels, "fb_maxw": maxw,
            "dft": dft.float().contiguous(), "win": win[:half + 1].contiguous(), "fb_start": start, "fb_count": count,
            "fb_w": fbw}


class MelSpectrogram(nn.Module):
    def __init__(self, sample_rate: int = 16000, n_fft: int = 400, win_length: Optional[int] = None,
                 hop_length: Optional[int] = None, f_min: float = 0.0, f_max: Optional[float] = None,
                 n_mels: int = 128) -> None:
        super().__init__()
        self.sample_rate, self.n_fft = sample_rate, n_fft
        self.win_length = n_fft if win_length is None else win_length
        self.hop_length = self.win_length // 2 if hop_length is None else hop_length
        self.f_min, self.f_max, self.n_mels = f_min, (float(sample_rate // 2) if f_max is None else f_max), n_mels
        plan = build_plan(sample_rate, n_fft, self.win_length, self.hop_length, n_mels, f_min, self.f_max)
        self._scalars = {k: v for k, v in plan.items() if not isinstance(v, torch.Tensor)}
        for k, v in plan.items():
            if isinstance(v, torch.Tensor):
                self.register_buffer("_" + k, v, persistent=False)       # derived tables: never in a state_dict

    def batch(self, waveforms: Tensor, lengths: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """(B, S) fp32 padded waveforms on the GPU, `lengths` (B,) samples per utterance (None: all S) ->
        ((B, T, n_mels) fp32 with T = 1 + S // hop_length and zeros past each utterance's frames, (B,) int64
        frame counts)."""
        if not waveforms.is_cuda:
            raise RuntimeError("conformer_amd: the mel front-end runs on MI355X through libfcenc_hip.so only "
                               "(there is no CPU execution path)")
        if self._dft.device != waveforms.device:
            raise RuntimeError(f"MelSpectrogram is on {self._dft.device}, the waveform on {waveforms.device}: call .to(device)")
        if waveforms.dim() != 2 or waveforms.dtype != torch.float32:
            raise ValueError(f"waveforms must be (batch, samples) fp32, got {tuple(waveforms.shape)} {waveforms.dtype}")
        b, s = waveforms.shape
        if s <= self.n_fft // 2:
            raise RuntimeError(f"reflect padding of {self.n_fft // 2} samples needs more than that many samples, got {s}")
        if waveforms.stride(1) != 1:
            waveforms = waveforms.contiguous()
        if lengths is not None:
            if not lengths.is_cuda and lengths.numel() and int(lengths.min()) <= self.n_fft // 2:
                # the check torch.stft makes per utterance; only possible without a device round trip when the
                # lengths are still on the host (device-resident lengths are not checked: see include/fcenc.h)
                raise RuntimeError(f"reflect padding of {self.n_fft // 2} samples needs utterances longer than that, "
                                   f"got one of {int(lengths.min())} samples")
            lengths = lengths.to(device=waveforms.device, dtype=torch.int64).contiguous()
            if lengths.numel() != b:
                raise ValueError("lengths must hold one sample count per utterance")
        from . import ops
        sc = self._scalars
        return ops.mel_spectrogram(waveforms, lengths, self._dft, self._win, self._fb_start, self._fb_count, self._fb_w,
                                   sc["n_fft"], sc["hop"], sc["kp"], sc["wpitch"], sc["n_mels"], sc["fb_maxw"])

    def forward(self, waveform: Tensor) -> Tensor:
        """(..., samples) -> (..., n_mels, frames) as torchaudio returns it (a transposed view of the
        (..., frames, n_mels) buffer the encoder reads, so the caller's `.transpose(-1, -2)` is contiguous)."""
        lead = waveform.shape[:-1]
        mel, _ = self.batch(waveform.reshape(-1, waveform.shape[-1]).float())
        return mel.reshape(lead + mel.shape[-2:]).transpose(-1, -2)
