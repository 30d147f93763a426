"""Mel front-end: the kernel against a torch composite of the same transform on the same GPU (torch.stft on
rocFFT + power + filter-bank matmul, what torchaudio's MelSpectrogram executes).

    python tools/mel_bench.py [--batch 64] [--seconds 20] [--n-fft 400] [--win 400]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd.frontend import MelSpectrogram, mel_filter_bank      # noqa: E402


def timeit(run, n=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        run()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--n-fft", type=int, default=400)
    ap.add_argument("--win", type=int, default=400)
    a = ap.parse_args()
    S = int(a.seconds * 16000)
    w = torch.randn(a.batch, S, device="cuda") * 0.1
    m = MelSpectrogram(16000, n_fft=a.n_fft, win_length=a.win, hop_length=160, n_mels=80).cuda()
    us = timeit(lambda: m.batch(w))
    frames = a.batch * (1 + S // 160)
    half = a.n_fft // 2
    from conformer_amd.frontend import plan_geometry
    geo = plan_geometry(a.n_fft)
    blocks = geo["b0"] + geo["b1"] + geo["b2"] + geo["b3"]
    flops = frames * geo["kp"] * blocks * 16 * 2          # MFMA work issued (twice-folded DFT incl. padding)
    window = torch.hann_window(a.win, device="cuda")
    fb = mel_filter_bank(half + 1, 0.0, 8000.0, 80, 16000).cuda()

    def composite():
        spec = torch.stft(w, a.n_fft, 160, a.win, window, center=True, pad_mode="reflect", return_complex=True)
        power = spec.real ** 2 + spec.imag ** 2
        return torch.matmul(power.transpose(1, 2), fb)

    ref_us = timeit(composite, 5)
    got, _ = m.batch(w)
    ref = composite()
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    print(json.dumps({"batch": a.batch, "seconds": a.seconds, "n_fft": a.n_fft, "win_length": a.win, "frames": frames,
                      "hip_us": round(us, 1), "hip_audio_s_per_s": round(a.batch * a.seconds / (us * 1e-6), 0),
                      "mfma_fp32_tflops": round(flops / us / 1e6, 1), "mfma_fp32_frac_of_157": round(flops / us / 1e6 / 157.3, 3),
                      "torch_stft_composite_us": round(ref_us, 1), "max_abs_diff_over_max": err}))


if __name__ == "__main__":
    main()
