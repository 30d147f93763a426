"""CPU oracle of the mel-spectrogram front-end the reference applies before the encoder
(train.py:57-61, 67-71 and inference.py:88-92: `torchaudio.transforms.MelSpectrogram(sample_rate, n_mels,
hop_length)` with n_fft = 400; inference.py:294-300: n_fft = 512, win_length = 400).

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path.

PARITY UNPINNED.  The algorithm lives in a third-party dependency, torchaudio (unpinned in the reference's
requirements.txt:9), which is not installed here and cannot be fetched, and the reference holds no fixture of
its output.  This file restates torchaudio's published algorithm for the default arguments the reference uses:

  MelSpectrogram = MelScale(Spectrogram(waveform))
  Spectrogram    : torch.stft(waveform, n_fft, hop_length, win_length, window = hann_window(win_length)
                   [periodic], center = True, pad_mode = "reflect", normalized = False, onesided = True),
                   then |.|^power with power = 2
  MelScale       : triangular filters melscale_fbanks(n_freqs = n_fft // 2 + 1, f_min = 0, f_max = sample_rate / 2,
                   n_mels, sample_rate, norm = None, mel_scale = "htk"); mel = fb^T . spectrogram

torch.stft itself is the real routine (it is part of torch, which is installed), so the DFT, the framing, the
reflect padding and the centring of a short window inside n_fft are pinned by torch; the filter bank is the
restated part.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

Tensor = torch.Tensor


def hz_to_mel_htk(f: float) -> float:
    """torchaudio.functional._hz_to_mel, mel_scale = "htk"."""
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") -> (n_freqs, n_mels) fp32."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]                              # (n_mels + 1)
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)         # (n_freqs, n_mels + 2)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def mel_spectrogram(waveform: Tensor, sample_rate: int = 16000, n_fft: int = 400, win_length: Optional[int] = None,
                    hop_length: Optional[int] = None, n_mels: int = 80) -> Tensor:
    """waveform (..., samples) fp32 -> (..., n_mels, 1 + samples // hop_length) fp32."""
    win_length = n_fft if win_length is None else win_length
    hop_length = win_length // 2 if hop_length is None else hop_length
    shape = waveform.shape
    w = waveform.reshape(-1, shape[-1]).float()
    spec = torch.stft(w, n_fft, hop_length, win_length, torch.hann_window(win_length), center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                      # (N, n_freqs, frames)
    fb = melscale_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate)
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)
    return mel.reshape(shape[:-1] + mel.shape[-2:])


def batch_mel(waveforms: Tensor, lengths: Optional[List[int]], **kw) -> Tuple[Tensor, Tensor]:
    """What the reference's dataset + collate produce for a batch (train.py:122 per utterance, then zero
    padding to the longest, train.py:170-174): (B, T_max, n_mels) fp32 and int64 frame counts."""
    hop = kw.get("hop_length") or (kw.get("win_length") or kw.get("n_fft", 400)) // 2
    B, S = waveforms.shape
    lengths = [S] * B if lengths is None else list(lengths)
    frames = [1 + n // hop for n in lengths]
    n_mels = kw.get("n_mels", 80)
    out = torch.zeros(B, max(frames), n_mels)
    for b in range(B):
        out[b, :frames[b]] = mel_spectrogram(waveforms[b, :lengths[b]], **kw).transpose(0, 1)
    return out, torch.tensor(frames, dtype=torch.int64)
