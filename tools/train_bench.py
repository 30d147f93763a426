"""Time of one training step of the encoder (forward + backward) on the HIP training path against ATen on the same GPU:
train.py:302-336 with the transducer head left out (loss = mean square of the encoder output), fp32 and AMP.

    python tools/train_bench.py [--batch 16] [--frames 1000] [--steps 10]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conformer_amd as fc                                          # noqa: E402
from conformer_amd.synth import make_state_dict, synthetic_lengths, synthetic_mel   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--only", default="", help="one of 'fp32-hip', 'fp32-aten', 'amp-hip', 'amp-aten' (for profiles)")
    a = ap.parse_args()
    cfg = fc.load_config(os.path.join(os.path.dirname(__file__), "..", "configs", "fastconformer_small.json"))
    cfg["model"]["dropout"] = 0.0
    cfg["model"]["stochastic_depth_prob"] = 0.0
    lens = synthetic_lengths(a.batch, a.frames, 1)
    x, ln = synthetic_mel(a.batch, a.frames, lens, 2)
    x, ln = x.cuda(), ln.cuda()
    res = {"batch": a.batch, "frames": a.frames, "config": "fastconformer_small, 17 layers"}
    for amp in (False, True):
        for native in (True, False):
            if a.only and a.only != ("amp-" if amp else "fp32-") + ("hip" if native else "aten"):
                continue
            m = fc.create_model(cfg)
            m.load_state_dict(make_state_dict(cfg, 3))
            m = m.cuda().train()
            m.encoder.native_training = native
            opt = torch.optim.AdamW(m.encoder.parameters(), lr=1e-4)

            def step():
                opt.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                    out, _, _ = m.encoder(x, ln)
                out.float().pow(2).mean().backward()
                opt.step()

            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            res[("amp " if amp else "fp32 ") + ("hip" if native else "aten")] = round(ms, 2)
            del m, opt
    print(json.dumps(res))


if __name__ == "__main__":
    main()
