#!/usr/bin/env python3
"""A/B of a front-end kernel choice inside whole forwards, same process, alternating.

    python tools/ab_frontend.py [--what conv12|conv3] [--config configs/fastconformer_small.json --batch 64 --frames 2000]

conv12: the fused conv1 -> conv2 kernel against the two kernels; conv3: the patch-in-LDS conv3 against the implicit GEMM."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conformer_amd as fc                                          # noqa: E402
from conformer_amd.synth import make_state_dict, synthetic_lengths, synthetic_mel   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="configs/fastconformer_small.json")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=2000)
ap.add_argument("--what", choices=("conv12", "conv3"), default="conv12")
a = ap.parse_args()
cfg = fc.load_config(a.config)
dev = torch.device("cuda:0")
m = fc.create_model(cfg)
m.load_state_dict(make_state_dict(cfg, 1234))
m.encoder.set_precision("bf16")
m = m.to(dev).eval()
lc = cfg["model"].get("local_context", 256)
t3 = a.frames
for _ in range(3):
    t3 = (t3 + 1) // 2
lens = synthetic_lengths(a.batch, a.frames, 100, min_frames=8 * max(0, t3 - lc + 1))
mel, ln = synthetic_mel(a.batch, a.frames, lens, 200)
mel, ln = mel.to(dev), ln.to(dev)
pk = m.encoder.packed(dev)
attr = {"conv12": "fuse_conv12", "conv3": "use_conv3_patch"}[a.what]
names = {"conv12": ("fused", "two_kernels"), "conv3": ("patch", "implicit_gemm")}[a.what]
res = {names[0]: [], names[1]: []}
with torch.inference_mode():
    for rnd in range(4):
        for name, flag in ((names[0], True), (names[1], False)):
            setattr(pk, attr, flag)
            for _ in range(3):
                m.encoder(mel, ln)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                m.encoder(mel, ln)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(round(e0.elapsed_time(e1) / 10, 4))
print(json.dumps({"what": a.what, "config": os.path.basename(a.config), "batch": a.batch, "frames": a.frames, "ms_per_forward": res}))
