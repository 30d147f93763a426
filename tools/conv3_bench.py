#!/usr/bin/env python3
"""Times the third subsampling convolution: the patch-in-LDS kernel against the implicit GEMM (bf16, 20 columns, 256 channels).

    python tools/conv3_bench.py [--batch 64 --rows 500 --iters 20]

Prints one JSON line: microseconds per call of either kernel (HIP events around `iters` back-to-back calls), the patch
kernel's share of the bf16 MFMA peak (2 * B*ceil(rows/2)*10*256*2304 FLOP) and the largest difference between them.
The input is larger than the caches and is read from HBM, as in the encoder pass."""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import _native as N                      # noqa: E402
from tools.conv12_bench import timed                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rows", type=int, default=500)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--zeros", action="store_true", help="all-zero input and weights: the same instructions with idle data paths")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = N.lib()
    b, h, c = args.batch, args.rows, 256
    g = torch.Generator().manual_seed(1)
    x = torch.randn((b, h, 20, c), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(c, 9 * c, generator=g) / math.sqrt(9 * c)).to(torch.bfloat16).to(dev)
    bias = (torch.randn(c, generator=g) * 0.1).to(dev)
    if args.zeros:
        x.zero_()
        w.zero_()
    hout = (h + 1) // 2
    hp = (hout + 63) // 64 * 64
    o_gemm = torch.zeros((b, hp, 10, c), dtype=torch.bfloat16, device=dev)
    o_patch = torch.zeros_like(o_gemm)
    wp = torch.empty(lib.fcenc_conv3_patch_packed_bytes(), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    N.check(lib.fcenc_conv3_patch_pack(w.data_ptr(), wp.data_ptr(), c, s))

    def gemm():
        N.check(lib.fcenc_conv3x3s2(N.BF16, x.data_ptr(), w.data_ptr(), bias.data_ptr(), o_gemm.data_ptr(), b, h, 20, c, c, hp, s))

    def patch():
        N.check(lib.fcenc_conv3_patch(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), o_patch.data_ptr(), b, h, 20, c, hp, s))

    us_g, us_p = timed(gemm, args.iters), timed(patch, args.iters)
    flops = 2.0 * b * hout * 10 * c * 9 * c
    diff = (o_patch.float() - o_gemm.float()).abs().max().item()
    print(json.dumps({"zeros": args.zeros, "batch": b, "rows": h, "implicit_gemm_us": round(us_g, 1), "patch_us": round(us_p, 1),
                      "patch_tflops": round(flops / us_p / 1e6, 1), "patch_frac_of_2500": round(flops / us_p / 1e6 / 2500, 4),
                      "input_GBps": round(x.numel() * 2 / us_p / 1e3, 1),
                      "max_abs_diff": diff, "out_max": o_gemm.float().abs().max().item()}))


if __name__ == "__main__":
    main()
