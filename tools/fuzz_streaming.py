"""Seeded random sweep of the cached streaming step against the CPU oracle, chunk by chunk: model widths, head
counts, depths, ring capacities (local_context), stream counts and chunk-length sequences drawn at random, with and
without HIP-graph replay and lanes, both precision modes.  One line per case; exits non-zero if any gate failed.

    python tools/fuzz_streaming.py [--cases 30] [--seed 1]

Gates per chunk: fp32 max-abs <= 1e-4; bf16 relative L2 <= 2e-2 and worst element <= 5e-2 of the largest.
"""
import argparse
import os
import random
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conformer_amd as fc                                          # noqa: E402
from conformer_amd.streaming import StreamCache                     # noqa: E402
from conformer_amd.synth import make_state_dict, synthetic_mel      # noqa: E402
from oracle import fc_oracle as O                                   # noqa: E402

WIDTHS = [(128, 2), (128, 4), (256, 4), (256, 8), (384, 8), (512, 8), (384, 6)]


def draw(rng: random.Random):
    d, h = rng.choice(WIDTHS)
    cfg = {"model": {"d_encoder": d, "encoder_heads": h, "encoder_layers": rng.choice([1, 2, 3]),
                     "subsampling_channels": rng.choice([128, 256]),
                     "local_context": rng.choice([8, 16, 24, 64, 256, 512])}}
    b = rng.choice([1, 2, 3, 5, 21, 22, 64, 65])
    base = rng.choice([17, 17, 9, 25, 33, 8, 1])
    n = rng.randint(3, 14)
    chunks = [base if rng.random() < 0.7 else rng.choice([1, 8, 9, 16, 17, 24, 25, 57, 64, 100, 128]) for _ in range(n)]
    # the HIP step takes chunks of at most min(16, local_context) encoder frames (longer ones run the composite)
    most = 8 * min(16, cfg["model"]["local_context"])
    chunks = [min(c, most) for c in chunks]
    return cfg, b, chunks, rng.random() < 0.3, rng.choice([0, 0, 2, 3])


def sweep(cases: int, seed: int, start: int = 0, inf_gate: float = 5e-2, emit=print):
    """Runs cases start .. cases-1 of the seeded sequence; returns (number of failed gates, worst errors seen)."""
    a = argparse.Namespace(cases=cases, seed=seed, start=start)
    return _sweep(a, inf_gate, emit)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--start", type=int, default=0)
    a = ap.parse_args()
    failures, _ = _sweep(a, 5e-2, print)
    sys.exit(1 if failures else 0)


def _sweep(a, inf_gate, emit):
    rng = random.Random(a.seed)
    failures = 0
    worst = {"fp32": 0.0, "bf16_l2": 0.0, "bf16_inf": 0.0}
    for case in range(a.cases):
        cfg, b, chunks, graph, lanes = draw(rng)
        if case < a.start:
            continue
        sd = make_state_dict(cfg, 3000 + case)
        x, _ = synthetic_mel(b, sum(chunks), None, 4000 + case)
        refs, cache, pos = [], None, 0
        for cf in chunks:
            out, _, cache = O.encoder_forward(sd, cfg, x[:, pos:pos + cf], None, cache=cache, streaming=True)
            refs.append(out)
            pos += cf
        line = f"case {case:3d} d={cfg['model']['d_encoder']} H={cfg['model']['encoder_heads']} L={cfg['model']['encoder_layers']} " \
               f"C={cfg['model']['subsampling_channels']} cap={cfg['model']['local_context']} B={b} chunks={chunks} graph={graph} lanes={lanes}"
        for precision in ("fp32", "bf16"):
            m = fc.create_model(cfg)
            m.load_state_dict(sd)
            m.encoder.set_precision(precision)
            m = m.cuda().eval()
            m.encoder.stream_graph = graph
            m.encoder.stream_lanes = lanes
            cache, pos, bad = None, 0, None
            e32 = l2w = infw = 0.0
            with torch.inference_mode(), warnings.catch_warnings():
                warnings.simplefilter("error", RuntimeWarning)        # a composite fallback would be reported here
                for i, cf in enumerate(chunks):
                    out, out_len, cache = m.encode(x[:, pos:pos + cf].cuda(), None, cache=cache, streaming=True)
                    pos += cf
                    if not isinstance(cache, StreamCache):
                        bad = f"chunk {i}: the HIP step did not run"
                        break
                    o, r = out.cpu(), refs[i]
                    if o.shape != r.shape or not torch.isfinite(o).all():
                        bad = f"chunk {i}: shape {tuple(o.shape)} vs {tuple(r.shape)} or not finite"
                        break
                    if r.numel() == 0:
                        continue
                    err = (o - r).abs().max().item()
                    if precision == "fp32":
                        e32 = max(e32, err)
                        if err > 1e-4:
                            bad = f"chunk {i}: fp32 max-abs {err:.2e}"
                            break
                    else:
                        l2, inf = ((o - r).norm() / r.norm()).item(), err / r.abs().max().item()
                        l2w, infw = max(l2w, l2), max(infw, inf)
                        if l2 > 2e-2 or inf > inf_gate:
                            bad = f"chunk {i}: bf16 l2 {l2:.2e} inf {inf:.2e}"
                            break
            if precision == "fp32":
                worst["fp32"] = max(worst["fp32"], e32)
                line += f" | fp32 {e32:.2e}"
            else:
                worst["bf16_l2"], worst["bf16_inf"] = max(worst["bf16_l2"], l2w), max(worst["bf16_inf"], infw)
                line += f" | bf16 l2 {l2w:.2e} inf {infw:.2e}"
            if bad:
                line += f"\n  GATE FAILED ({precision}): {bad}"
                failures += 1
            del m
        emit(line)
    emit(f"worst: {worst} failures: {failures}")
    return failures, worst


if __name__ == "__main__":
    main()
