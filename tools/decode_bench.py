"""Greedy RNN-T decode of a whole encoded batch: the device kernel (`fcenc_greedy_decode`, one launch)
against the reference's per-symbol Python loop (train.py:180-211) run with the same PyTorch modules on
the same GPU, on a bounded sample of utterances.

    python tools/decode_bench.py [--batch 64] [--frames 2000] [--max-symbols 10] [--loop-utts 1]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conformer_amd as fc                                                        # noqa: E402
from conformer_amd import decode as D                                             # noqa: E402
from conformer_amd.synth import make_state_dict, synthetic_lengths, synthetic_mel  # noqa: E402


def python_loop(model, enc, max_symbols):
    """The reference's loop, statement for statement in behaviour (one .item() per decision)."""
    out = []
    for b in range(enc.shape[0]):
        toks = []
        po = model.predict(torch.zeros(1, 1, dtype=torch.long, device=enc.device))
        for t in range(enc.shape[1]):
            e = enc[b:b + 1, t:t + 1]
            for _ in range(max_symbols):
                p = model.joint_step(e, po).argmax(dim=-1).item()
                if p == model.blank_id:
                    break
                toks.append(p)
                po = model.predict(torch.tensor([[p]], dtype=torch.long, device=enc.device))
        out.append(toks)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--max-symbols", type=int, default=10)
    ap.add_argument("--loop-utts", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    config = fc.load_config(os.path.join(os.path.dirname(__file__), "..", "configs", "fastconformer_small.json"))
    m = fc.create_model(config)
    m.load_state_dict(make_state_dict(config, 1234))
    m.encoder.set_precision("bf16")
    m = m.cuda().eval()
    lens = synthetic_lengths(a.batch, a.frames, 100)
    x, ln = synthetic_mel(a.batch, a.frames, lens, 200)
    with torch.inference_mode():
        enc, enc_len, _ = m.encode(x.cuda(), ln.cuda())
        tokens, counts, _ = D.greedy_decode_device(m, enc, a.max_symbols)     # builds the tables
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(a.steps):
            tokens, counts, _ = D.greedy_decode_device(m, enc, a.max_symbols)
        ev1.record()
        torch.cuda.synchronize()
        dev_ms = ev0.elapsed_time(ev1) / a.steps
        n_sym = int(counts.sum())
        decisions = n_sym + a.batch * enc.shape[1]      # upper bound: every frame ends with one more decision unless capped
        res = {"batch": a.batch, "encoder_frames": enc.shape[1], "max_symbols": a.max_symbols, "symbols": n_sym,
               "device_ms": round(dev_ms, 3), "device_us_per_decision": round(dev_ms * 1e3 * a.batch / decisions, 3)}
        if a.loop_utts > 0:
            t0 = time.perf_counter()
            ref = python_loop(m, enc[:a.loop_utts], a.max_symbols)
            torch.cuda.synchronize()
            loop_s = time.perf_counter() - t0
            c = counts.cpu().tolist()
            got = [tokens[i, :c[i]].cpu().tolist() for i in range(a.loop_utts)]
            res["python_loop_s_per_utterance"] = round(loop_s / a.loop_utts, 3)
            res["python_loop_ms_whole_batch_extrapolated"] = round(loop_s / a.loop_utts * a.batch * 1e3, 1)
            res["same_tokens_as_loop"] = got == ref
    print(json.dumps(res))


if __name__ == "__main__":
    main()
