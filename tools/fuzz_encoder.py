"""Seeded random sweep of the offline encoder against the CPU oracle: model widths, head counts, depths,
subsampling widths, window settings, batch sizes, frame counts and length patterns drawn at random, both
precision modes.  Prints one line per case (per-utterance errors where a gate fails) and a summary; exits non-zero if any gate failed.

    python tools/fuzz_encoder.py [--cases 40] [--seed 1]

Gates (the ones tests/test_gpu_encoder.py applies): output lengths bit-exact; over the valid frames of every
utterance fp32 max-abs <= 1e-4, bf16 relative L2 <= 2e-2 and worst element <= 8e-2 of the largest (1-3-block models of the sweep).
Frames past an utterance's length are not compared (the reference leaves NaN there when a causal window holds no
valid key).
"""
import argparse
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conformer_amd as fc                                          # noqa: E402
from conformer_amd.synth import make_state_dict, synthetic_mel      # noqa: E402
from oracle import fc_oracle as O                                   # noqa: E402

WIDTHS = [(128, 2), (128, 4), (256, 4), (256, 8), (384, 8), (512, 8), (384, 6)]   # (d_encoder, heads): head_dim 64, 32, 64, 32, 48, 64, 64


def draw(rng: random.Random):
    d, h = rng.choice(WIDTHS)
    cfg = {"model": {"d_encoder": d, "encoder_heads": h, "encoder_layers": rng.choice([1, 2, 3]),
                     "subsampling_channels": rng.choice([128, 256]),
                     "use_local_attn": rng.random() < 0.8, "local_context": rng.choice([8, 16, 40, 64, 256])}}
    b = rng.choice([1, 1, 2, 3, 5, 7])
    t = rng.choice([8, 9, 15, 16, 17, 63, 64, 65, 100, 257, 511, 512, 513, 700, 1000, rng.randint(8, 1200)])
    kind = rng.choice(["none", "full", "ragged", "short"])
    if kind == "none":
        lens = None
    elif kind == "full":
        lens = [t] * b
    elif kind == "ragged":
        lens = [t] + [rng.randint(max(1, t // 3), t) for _ in range(b - 1)]
    else:
        lens = [t] + [rng.randint(1, min(t, 24)) for _ in range(b - 1)]
    if lens is not None:
        rng.shuffle(lens)
        if max(lens) != t:
            lens[0] = t
    return cfg, b, t, lens


def sweep(cases: int, seed: int, start: int = 0, inf_gate: float = 8e-2, emit=print):
    """Runs cases start .. cases-1 of the seeded sequence; returns (number of failed gates, worst errors seen)."""
    a = argparse.Namespace(cases=cases, seed=seed, start=start)
    return _sweep(a, inf_gate, emit)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--start", type=int, default=0, help="skip the cases before this one (same draws)")
    a = ap.parse_args()
    failures, _ = _sweep(a, 8e-2, print)
    sys.exit(1 if failures else 0)


def _sweep(a, inf_gate, emit):
    rng = random.Random(a.seed)
    worst = {"fp32": 0.0, "bf16_l2": 0.0, "bf16_inf": 0.0}
    failures = 0
    for case in range(a.cases):
        cfg, b, t, lens = draw(rng)
        if case < a.start:
            continue
        sd = make_state_dict(cfg, 1000 + case)
        x, ln = synthetic_mel(b, t, lens, 2000 + case)
        ref, ref_len, _ = O.encoder_forward(sd, cfg, x, ln)
        line = f"case {case:3d} d={cfg['model']['d_encoder']} H={cfg['model']['encoder_heads']} L={cfg['model']['encoder_layers']} " \
               f"C={cfg['model']['subsampling_channels']} local={cfg['model']['use_local_attn']}/{cfg['model']['local_context']} " \
               f"B={b} T={t} lens={lens}"
        for precision in ("fp32", "bf16"):
            m = fc.create_model(cfg)
            m.load_state_dict(sd)
            m.encoder.set_precision(precision)
            m = m.cuda().eval()
            with torch.inference_mode():
                out, out_len, _ = m.encode(x.cuda(), None if ln is None else ln.cuda())
            out, out_len = out.cpu(), out_len.cpu()
            if not torch.equal(out_len, ref_len):
                emit(line + f"\n  {precision}: LENGTHS {out_len.tolist()} != {ref_len.tolist()}")
                return failures + 1, worst
            keep = torch.arange(ref.shape[1])[None, :] < ref_len[:, None]
            o, r = out[keep], ref[keep]
            if not torch.isfinite(r).all():
                line += " [reference not finite on valid frames: skipped]"
                break
            if not torch.isfinite(o).all():
                emit(line + f"\n  {precision}: output not finite on valid frames")
                return failures + 1, worst
            err = (o - r).abs().max().item()
            if precision == "fp32":
                worst["fp32"] = max(worst["fp32"], err)
                line += f" | fp32 {err:.2e}"
                ok = err <= 1e-4
            else:
                l2 = ((o - r).norm() / r.norm()).item()
                inf = err / r.abs().max().item()
                worst["bf16_l2"], worst["bf16_inf"] = max(worst["bf16_l2"], l2), max(worst["bf16_inf"], inf)
                line += f" | bf16 l2 {l2:.2e} inf {inf:.2e}"
                ok = l2 <= 2e-2 and inf <= inf_gate
            if not ok:
                per = []
                for u in range(b):
                    n = int(ref_len[u])
                    per.append(f"utt {u} ({n} frames): l2 {((out[u, :n] - ref[u, :n]).norm() / ref[u, :n].norm()).item():.2e} "
                               f"max {(out[u, :n] - ref[u, :n]).abs().max().item():.2e}")
                line += "\n  GATE FAILED: " + "; ".join(per)
                failures += 1
            del m
        emit(line)
    emit(f"worst: {worst} failures: {failures}")
    return failures, worst


if __name__ == "__main__":
    main()
