"""Micro-benchmark of the streaming attention kernel alone (saturated caches).

    python tools/stream_attn_bench.py [--streams 1024] [--heads 4] [--cap 256] [--tq 3] [--sets 6]

Rotates over `--sets` independent ring sets so that no call finds its caches in the memory-side cache.
Reports microseconds per call and the rate at which the algorithmic bytes (K and V rings once) move.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import ops      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--cap", type=int, default=256)
    ap.add_argument("--tq", type=int, default=3)
    ap.add_argument("--sets", type=int, default=6)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    b, h, cap, tq = a.streams, a.heads, a.cap, a.tq
    hd = h * 64
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = (torch.randn((b * tq, 3 * hd), device=dev, generator=g) * 0.5).bfloat16()
    pos = (torch.randn((2 * tq - 1, hd), device=dev, generator=g) * 0.5).bfloat16()
    bu = torch.randn(hd, device=dev, generator=g) * 0.1
    bv = torch.randn(hd, device=dev, generator=g) * 0.1
    sets = [((torch.randn((b, h, cap, 64), device=dev, generator=g) * 0.5).bfloat16(),
             (torch.randn((b, h, cap // 8, 64, 8), device=dev, generator=g) * 0.5).bfloat16()) for _ in range(a.sets)]
    frames = 10 * cap
    for kc, vc in sets:
        ops.stream_attention_(qkv, pos, bu, bv, kc, vc, frames, 0.125)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(a.iters):
        for kc, vc in sets:
            ops.stream_attention_(qkv, pos, bu, bv, kc, vc, frames + tq * it, 0.125)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (a.iters * a.sets)
    nbytes = b * h * 2 * cap * 64 * 2
    res = {"us_per_call": round(us, 2), "ring_bytes": nbytes, "GB_per_s": round(nbytes / us * 1e-3, 1)}
    # what the same bytes cost as a plain device copy (read + write) and as a read-only reduction
    src = [s[0] for s in sets]
    dst = torch.empty_like(src[0])
    e0.record()
    for it in range(a.iters):
        for s in src:
            dst.copy_(s)
    e1.record()
    torch.cuda.synchronize()
    us_c = e0.elapsed_time(e1) * 1e3 / (a.iters * a.sets)
    res["copy_GB_per_s_read_plus_write"] = round(2 * src[0].numel() * 2 / us_c * 1e-3, 1)
    e0.record()
    for it in range(a.iters):
        for s in src:
            s.view(torch.int32).max()
    e1.record()
    torch.cuda.synchronize()
    us_r = e0.elapsed_time(e1) * 1e3 / (a.iters * a.sets)
    res["reduce_GB_per_s_read"] = round(src[0].numel() * 2 / us_r * 1e-3, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
