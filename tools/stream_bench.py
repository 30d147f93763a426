"""Chunk latency of the cached streaming encoder: `--streams` concurrent streams, chunks of `--frames`
mel frames (17 = 160 ms + lookahead of the serving configuration), fastconformer_small bf16.

    python tools/stream_bench.py [--streams 1024] [--frames 17] [--warm 100] [--steps 50] [--composite]

The caches are saturated first (`--warm` chunks; local_context 256 needs 86 chunks of 3 frames).
`--composite` also times the PyTorch composite with the reference's cat / slice caches on the same GPU.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conformer_amd as fc                                          # noqa: E402
from conformer_amd.synth import make_state_dict, synthetic_mel     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=17)
    ap.add_argument("--warm", type=int, default=100)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--composite", action="store_true")
    ap.add_argument("--lanes", type=int, default=0, help="concurrent slices of the batch (0 = library default)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph")
    ap.add_argument("--sync-every", type=int, default=0, help="synchronise the device every N warm-up chunks "
                    "(bounds how many launches the host queues ahead; 0 = never)")
    a = ap.parse_args()
    config = {"model": {"d_encoder": 256, "encoder_layers": 17, "encoder_heads": 4, "local_context": 256}}
    m = fc.create_model(config)
    m.load_state_dict(make_state_dict(config, 1))
    m.encoder.set_precision(a.precision)
    m = m.cuda().eval()
    m.encoder.stream_lanes = a.lanes
    m.encoder.stream_graph = a.graph
    x, _ = synthetic_mel(a.streams, a.frames, None, 2)
    x = x.cuda()

    def run(encode, n_warm, n_steps):
        cache = None
        with torch.inference_mode():
            for i in range(n_warm):
                _, _, cache = encode(x, cache)
                if a.sync_every and (i + 1) % a.sync_every == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_steps):
                _, _, cache = encode(x, cache)
            host = time.perf_counter() - t0          # the host is done queueing here
            torch.cuda.synchronize()
        run.host_ms = host / n_steps * 1e3
        return (time.perf_counter() - t0) / n_steps * 1e3

    res = {"streams": a.streams, "chunk_mel_frames": a.frames, "precision": a.precision, "lanes": a.lanes, "graph": a.graph}
    ms = run(lambda c, cache: m.encode(c, None, cache=cache, streaming=True), a.warm, a.steps)
    res["hip_ms_per_chunk"] = round(ms, 4)
    res["host_queueing_ms_per_chunk"] = round(run.host_ms, 4)
    res["hip_audio_s_per_s"] = round(a.streams * a.frames * 0.01 / (ms * 1e-3), 1)
    if a.composite:
        enc = m.encoder
        ms = run(lambda c, cache: enc.forward_composite(c, None, cache, True), a.warm, max(a.steps // 5, 5))
        res["composite_ms_per_chunk"] = round(ms, 4)
        res["composite_audio_s_per_s"] = round(a.streams * a.frames * 0.01 / (ms * 1e-3), 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
