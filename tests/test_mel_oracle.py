"""CPU checks of the mel front-end's oracle and of the host-built plan the kernel consumes.

torchaudio is not installed, so the oracle restates its published algorithm (parity unpinned against
torchaudio itself); what can be pinned here is pinned: the oracle against the textbook definition evaluated
directly in fp64 (framing, reflect padding, window centring, one-sided power spectrum), the filter bank's
defining properties, and the folded DFT matrix + sparse filter bank of `conformer_amd.frontend.build_plan`
against the oracle."""
import math

import pytest
import torch

from conformer_amd.frontend import build_plan, plan_geometry
from oracle import mel_oracle as MO


def _direct(w, n_fft, win_length, hop, n_mels, sr=16000):
    """Definition, no FFT, fp64."""
    pad = n_fft // 2
    idx = torch.arange(-pad, w.numel() + pad)
    idx = torch.where(idx < 0, -idx, idx)
    idx = torch.where(idx >= w.numel(), 2 * (w.numel() - 1) - idx, idx)
    wp = w.double()[idx]
    lo = (n_fft - win_length) // 2
    win = torch.zeros(n_fft, dtype=torch.float64)
    n = torch.arange(win_length, dtype=torch.float64)
    win[lo:lo + win_length] = 0.5 - 0.5 * torch.cos(2 * math.pi * n / win_length)
    frames = 1 + w.numel() // hop
    k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)[:, None]
    nn_ = torch.arange(n_fft, dtype=torch.float64)[None, :]
    c, s = torch.cos(2 * math.pi * k * nn_ / n_fft), torch.sin(2 * math.pi * k * nn_ / n_fft)
    fb = MO.melscale_fbanks(n_fft // 2 + 1, 0.0, sr / 2, n_mels, sr).double()
    out = torch.zeros(n_mels, frames, dtype=torch.float64)
    for t in range(frames):
        x = wp[t * hop:t * hop + n_fft] * win
        out[:, t] = ((c @ x) ** 2 + (s @ x) ** 2) @ fb
    return out


@pytest.mark.parametrize("n_fft,win", [(400, 400), (512, 400)])
def test_oracle_equals_the_definition(n_fft, win):
    g = torch.Generator().manual_seed(n_fft)
    w = torch.randn(3217, generator=g) * 0.3
    ref = _direct(w, n_fft, win, 160, 80)
    got = MO.mel_spectrogram(w, 16000, n_fft, win, 160, 80)
    assert got.shape == (80, 1 + 3217 // 160)
    assert (got.double() - ref).abs().max() <= 2e-5 * ref.abs().max()


def test_filter_bank_properties():
    fb = MO.melscale_fbanks(201, 0.0, 8000.0, 80, 16000)
    assert fb.shape == (201, 80) and (fb >= 0).all() and fb.max() <= 1.0
    peaks = fb.argmax(0)
    assert (peaks[1:] >= peaks[:-1]).all()                          # centres rise with the filter index
    # HTK scale: filter centres are equally spaced in mel
    mel_pts = torch.linspace(0.0, 2595.0 * math.log10(1 + 8000 / 700), 82)[1:-1]
    centre_hz = 700.0 * (10 ** (mel_pts / 2595.0) - 1)
    assert ((peaks.float() * 40.0 - centre_hz).abs() <= 40.0).all()  # within one 40 Hz bin of the centre
    # neighbouring triangles overlap to a partition of unity between the first and the last centre
    inner = fb.sum(1)[int(peaks[0]) + 1:int(peaks[-1])]
    assert (inner - 1.0).abs().max() < 1e-4


def test_tone_lands_in_its_filter():
    t = torch.arange(16000) / 16000.0
    for hz in (440.0, 1000.0, 3000.0):
        mel = MO.mel_spectrogram(torch.sin(2 * math.pi * hz * t), 16000, 400, 400, 160, 80)
        j = int(mel[:, 10:90].mean(1).argmax())
        fb = MO.melscale_fbanks(201, 0.0, 8000.0, 80, 16000)
        assert fb[int(round(hz / 40.0)), j] > 0.3


@pytest.mark.parametrize("n_fft,win", [(400, 400), (512, 400)])
def test_plan_reproduces_the_oracle(n_fft, win):
    """The kernel's arithmetic (window on the twice-folded samples, four cos / sin matrices, sparse filters)
    evaluated in fp64 on the CPU."""
    hop = 160
    p = build_plan(16000, n_fft, win, hop, 80, 0.0, 8000.0)
    geo = plan_geometry(n_fft)
    g = torch.Generator().manual_seed(5)
    w = torch.randn(2400, generator=g) * 0.1
    ref = MO.mel_spectrogram(w, 16000, n_fft, win, hop, 80)
    half, pad, q, kp = n_fft // 2, n_fft // 2, geo["q"], geo["kp"]
    wp = torch.nn.functional.pad(w[None, None], (pad, pad), mode="reflect")[0, 0].double()
    dft, wt = p["dft"].double(), p["win"].double()
    assert dft.shape == (kp, p["wpitch"]) and p["wpitch"] % 64 in (16, 48) and wt.shape == (half + 1,)
    o1 = 16 * geo["b0"]
    o2 = o1 + 16 * geo["b1"]
    o3 = o2 + 16 * geo["b2"]
    m = torch.arange(kp)
    first = (m > 0).double()                                        # x[N] and the second x[H] are left out at m = 0
    out = torch.zeros(ref.shape[1], 80, dtype=torch.float64)
    for t in range(ref.shape[1]):
        x = torch.cat([wp[t * hop:t * hop + n_fft], torch.zeros(1, dtype=torch.float64)])
        xa, xb, xc, xd = x[m], x[n_fft - m] * first, x[half - m], x[half + m] * first
        u, u2 = wt[m] * (xa + xb), wt[half - m] * (xc + xd)
        v, v2 = wt[m] * (xa - xb), wt[half - m] * (xc - xd)
        pw = torch.zeros(half + 1, dtype=torch.float64)
        pw[0::2] = ((u + u2) @ dft[:, :q + 1]) ** 2
        pw[1::2] = ((u - u2) @ dft[:, o1:o1 + q]) ** 2
        pw[2:half:2] += ((v - v2) @ dft[:, o2:o2 + q - 1]) ** 2
        pw[1::2] += ((v + v2) @ dft[:, o3:o3 + q]) ** 2
        for j in range(80):
            s, c = int(p["fb_start"][j]), int(p["fb_count"][j])
            out[t, j] = (p["fb_w"][j, :c].double() * pw[s:s + c]).sum()
    assert (out.T - ref.double()).abs().max() <= 2e-6 * ref.abs().max()


def test_plan_rejects_what_is_not_built():
    with pytest.raises(NotImplementedError):
        build_plan(16000, 1024, 1024, 160, 80, 0.0, 8000.0)
    with pytest.raises(ValueError):
        build_plan(16000, 400, 401, 160, 80, 0.0, 8000.0)


def test_front_end_has_no_cpu_path_and_no_state_dict_entries():
    """The product module refuses CPU tensors (the HIP kernel is the only execution path) and registers its derived
    tables as non-persistent buffers, so a model that holds it saves nothing new."""
    from conformer_amd import MelSpectrogram
    m = MelSpectrogram(16000, n_mels=80, hop_length=160)
    assert len(m.state_dict()) == 0
    assert m.n_fft == 400 and m.win_length == 400 and m.hop_length == 160 and m.f_max == 8000.0
    with pytest.raises(RuntimeError, match="no CPU execution path"):
        m(torch.zeros(1, 16000))
    m2 = MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80)
    assert m2._dft.shape == (132, 528) and m._dft.shape == (104, 464) and m2._win.shape == (257,)
