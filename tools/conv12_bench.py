#!/usr/bin/env python3
"""Times the fused conv1 -> conv2 kernel against the two kernels it replaces (bf16, 80 mel bins, 256 channels).

    python tools/conv12_bench.py [--batch 64 --frames 2000 --iters 20]

Prints one JSON line: microseconds per call of conv1, conv2 (conv256 kernel) and the fused kernel, each timed with HIP
events around `iters` back-to-back calls, and the fused kernel's share of the bf16 MFMA peak (2 * B*T2*20*256*2304 FLOP)."""
import argparse
import ctypes as C
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import _native as N                      # noqa: E402
from conformer_amd.synth import synthetic_lengths, synthetic_mel   # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = N.lib()
    b, t, c = args.batch, args.frames, 256
    g = torch.Generator().manual_seed(1)
    mel, _ = synthetic_mel(b, t, synthetic_lengths(b, t, 3), 4)
    mel = mel.to(dev)
    w9 = (torch.randn(9, c, generator=g) * 0.3).to(dev)
    b1 = (torch.randn(c, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(c, 9 * c, generator=g) / math.sqrt(9 * c)).to(torch.bfloat16).to(dev)
    b2 = (torch.randn(c, generator=g) * 0.1).to(dev)
    t1, t2 = (t + 1) // 2, ((t + 1) // 2 + 1) // 2
    partial = torch.empty((b, 16, 2), dtype=torch.float64, device=dev)
    c1 = torch.empty((b, t1, 40, c), dtype=torch.bfloat16, device=dev)
    o2 = torch.empty((b, t2, 20, c), dtype=torch.bfloat16, device=dev)
    o12 = torch.empty_like(o2)
    a1p = torch.empty(lib.fcenc_conv12_packed_bytes(0), dtype=torch.uint8, device=dev)
    w2p = torch.empty(lib.fcenc_conv12_packed_bytes(1), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    N.check(lib.fcenc_conv12_pack(w9.data_ptr(), b1.data_ptr(), w2.data_ptr(), a1p.data_ptr(), w2p.data_ptr(), 80, c, s))
    N.check(lib.fcenc_input_stats(mel.data_ptr(), partial.data_ptr(), b, t, 80, 16, s))

    def conv1():
        N.check(lib.fcenc_conv1(N.BF16, mel.data_ptr(), partial.data_ptr(), 16, w9.data_ptr(), b1.data_ptr(), c1.data_ptr(),
                                b, t, 80, c, s))

    def conv2():
        N.check(lib.fcenc_conv3x3s2(N.BF16, c1.data_ptr(), w2.data_ptr(), b2.data_ptr(), o2.data_ptr(), b, t1, 40, c, c, t2, s))

    def fused():
        N.check(lib.fcenc_lognorm_conv12(mel.data_ptr(), partial.data_ptr(), 16, a1p.data_ptr(), w2p.data_ptr(),
                                         b2.data_ptr(), o12.data_ptr(), b, t, 80, c, s))

    us1, us2, us12 = timed(conv1, args.iters), timed(conv2, args.iters), timed(fused, args.iters)
    flops = 2.0 * b * t2 * 20 * c * 9 * c
    diff = (o12.float() - o2.float()).abs().max().item()
    print(json.dumps({"batch": b, "frames": t, "conv1_us": round(us1, 1), "conv2_us": round(us2, 1),
                      "two_kernels_us": round(us1 + us2, 1), "fused_us": round(us12, 1),
                      "fused_tflops": round(flops / us12 / 1e6, 1), "fused_frac_of_2500": round(flops / us12 / 1e6 / 2500, 4),
                      "max_abs_diff_fused_vs_two": diff, "out_max": o2.float().abs().max().item()}))


if __name__ == "__main__":
    main()
