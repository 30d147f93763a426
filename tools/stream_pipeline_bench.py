"""The reference's streaming loop (inference.py:322-376) for many concurrent streams, every stage on the device:
one 160 ms waveform chunk per stream -> mel kernel (n_fft 512 / win 400) -> cached encoder step -> incremental
greedy decode.  Reports the time per chunk with the caches saturated.

    python tools/stream_pipeline_bench.py [--streams 1024] [--chunk-ms 160] [--warm 100] [--steps 50]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conformer_amd as fc                                          # noqa: E402
from conformer_amd import decode as D                               # noqa: E402
from conformer_amd.synth import make_state_dict                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--chunk-ms", type=int, default=160)
    ap.add_argument("--warm", type=int, default=100)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--lanes", type=int, default=0, help="lanes of the encoder step (0 = library default)")
    a = ap.parse_args()
    cfg = fc.load_config(os.path.join(os.path.dirname(__file__), "..", "configs", "fastconformer_small.json"))
    model = fc.create_model(cfg)
    model.load_state_dict(make_state_dict(cfg, 1))
    model.encoder.set_precision("bf16")
    model = model.cuda().eval()
    model.encoder.stream_lanes = a.lanes
    mel = fc.MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80).cuda()
    dec = D.IncrementalGreedyDecoder(model, a.streams, max_symbols_per_step=10)
    samples = 16 * a.chunk_ms
    g = torch.Generator().manual_seed(4)
    chunk = (torch.randn(a.streams, samples, generator=g) * 0.1).cuda()
    cache = None

    def step(ev=None):
        nonlocal cache
        if ev:
            ev[0].record()
        x, _ = mel.batch(chunk)
        if ev:
            ev[1].record()
        enc, _, cache = model.encode(x, None, cache=cache, streaming=True)
        if ev:
            ev[2].record()
        tokens, counts = dec.step_device(enc)
        if ev:
            ev[3].record()
        return x, enc, counts

    with torch.inference_mode():
        for _ in range(a.warm):
            step()
        torch.cuda.synchronize()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(a.steps)]
        for ev in evs:
            x, enc, counts = step(ev)
        torch.cuda.synchronize()
    avg = [sum(e[i].elapsed_time(e[i + 1]) for e in evs) / len(evs) for i in range(3)]
    total = sum(e[0].elapsed_time(e[3]) for e in evs) / len(evs)
    print(json.dumps({"streams": a.streams, "lanes": a.lanes, "chunk_ms": a.chunk_ms, "mel_frames": x.shape[1], "encoder_frames": enc.shape[1],
                      "mel_ms": round(avg[0], 3), "encoder_ms": round(avg[1], 3), "decode_ms": round(avg[2], 3),
                      "total_ms_per_chunk": round(total, 3),
                      "audio_s_per_s": round(a.streams * a.chunk_ms * 1e-3 / (total * 1e-3), 0),
                      "real_time_streams_at_this_rate": int(a.streams * a.chunk_ms / total),
                      "symbols_per_chunk": int(counts.sum())}))


if __name__ == "__main__":
    main()
