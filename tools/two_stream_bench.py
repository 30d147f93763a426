"""Experiment: one forward over 64 utterances on one stream vs concurrent forwards over slices of the batch on as many
streams.  `join` as second argument: the streams fork from and join the main stream at every step (what a single
library call with lanes would do); without it they run ahead independently and drift out of phase."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import conformer_amd as fc
from conformer_amd.synth import make_state_dict, synthetic_lengths, synthetic_mel

dev = torch.device("cuda:0")
cfg = fc.load_config(os.path.join(ROOT, "configs", "fastconformer_small.json"))
if len(sys.argv) > 1 and int(sys.argv[1]) > 0: cfg["model"]["encoder_layers"] = int(sys.argv[1])
model = fc.create_model(cfg); model.load_state_dict(make_state_dict(cfg, 1234)); model.encoder.set_precision("bf16")
model = model.to(dev).eval()
B, T = 64, 2000
lens = synthetic_lengths(B, T, 100); mel, ln = synthetic_mel(B, T, lens, 200); mel, ln = mel.to(dev), ln.to(dev)
pk = model.encoder.packed(dev)

JOIN = len(sys.argv) > 2 and sys.argv[2] == "join"


def run(nsplit, steps=20):
    streams = [torch.cuda.Stream(dev) for _ in range(nsplit)]
    wss = [{} for _ in range(nsplit)]
    parts = [(mel[i * B // nsplit:(i + 1) * B // nsplit].contiguous(), ln[i * B // nsplit:(i + 1) * B // nsplit].contiguous()) for i in range(nsplit)]
    main = torch.cuda.current_stream(dev)

    def step():
        outs = []
        if JOIN:
            fork = torch.cuda.Event()
            fork.record(main)
        for s, ws, (m, l) in zip(streams, wss, parts):
            if JOIN:
                s.wait_event(fork)
            with torch.cuda.stream(s):
                outs.append(pk.forward(m, l, ws)[0])
            if JOIN:
                e = torch.cuda.Event()
                e.record(s)
                main.wait_event(e)
        return outs
    with torch.inference_mode():
        for _ in range(3): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): step()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for n in (1, 2, 4, 1, 2, 4, 3):
    print(f"splits={n}: {run(n):.3f} ms/step", flush=True)
