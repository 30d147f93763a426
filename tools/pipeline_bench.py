"""Whole offline pipeline on one GPU: padded waveforms -> mel kernel -> encoder forward -> greedy decode kernel,
fastconformer_small bf16, the benchmark batch (64 utterances x 20 s, ragged).  Every stage is queued on the same
stream without a host synchronisation in between; reported per stage (HIP events) and end to end.

    python tools/pipeline_bench.py [--batch 64] [--seconds 20] [--max-symbols 10] [--steps 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conformer_amd as fc                                                      # noqa: E402
from conformer_amd import decode as D                                           # noqa: E402
from conformer_amd.synth import make_state_dict                                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--max-symbols", type=int, default=10)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    cfg = fc.load_config(os.path.join(os.path.dirname(__file__), "..", "configs", "fastconformer_small.json"))
    model = fc.create_model(cfg)
    model.load_state_dict(make_state_dict(cfg, 1234))
    model.encoder.set_precision("bf16")
    model = model.cuda().eval()
    mel = fc.MelSpectrogram(16000, n_mels=80, hop_length=160).cuda()
    S = int(a.seconds * 16000) - 160                 # 1 + S // 160 frames = 100 per second, a multiple of 8
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(S // 2, S + 1, (a.batch,), generator=g)
    lens[0] = S
    wave = (torch.randn(a.batch, S, generator=g) * 0.1).cuda()
    for b in range(a.batch):
        wave[b, int(lens[b]):] = 0.0
    lens = lens.cuda()

    def step(ev=None):
        if ev:
            ev[0].record()
        x, frames = mel.batch(wave, lens)
        if ev:
            ev[1].record()
        enc, enc_len, _ = model.encode(x, frames)
        if ev:
            ev[2].record()
        tokens, counts, _ = D.greedy_decode_device(model, enc, a.max_symbols)
        if ev:
            ev[3].record()
        return tokens, counts

    with torch.inference_mode():
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(a.steps)]
        for ev in evs:
            tokens, counts = step(ev)
        torch.cuda.synchronize()
    ms = [[e[i].elapsed_time(e[i + 1]) for i in range(3)] for e in evs]
    avg = [sum(m[i] for m in ms) / len(ms) for i in range(3)]
    total = sum(e[0].elapsed_time(e[3]) for e in evs) / len(evs)
    audio = float(lens.sum()) / 16000.0
    print(json.dumps({"batch": a.batch, "seconds": a.seconds, "real_audio_s": round(audio, 1),
                      "mel_ms": round(avg[0], 3), "encoder_ms": round(avg[1], 3), "decode_ms": round(avg[2], 3),
                      "total_ms": round(total, 3), "audio_s_per_s": round(audio / (total * 1e-3), 0),
                      "symbols": int(counts.sum()), "max_symbols": a.max_symbols}))


if __name__ == "__main__":
    main()
