#!/usr/bin/env python3
"""Headline benchmark: FastConformer-small encoder forward, bf16, batch 64 x 20 s per GPU.

    python bench.py [--gpus N --steps K --warmup W]            (N > 1: launched by torch.distributed.run)

One step = one encoder forward (libfcenc_hip.so, one C call) over a batch of synthetic
utterances already resident in HBM, followed for N > 1 by the RCCL all-gather of outputs and
lengths (started asynchronously: it overlaps the next step's forward).  Prints ONE JSON line on rank 0.  `value` = real (unpadded) audio seconds encoded per
wall second by the whole job; the padded figure is in `config`.  `roofline` prices the dominant
kernel (conv2 of the subsampling front-end, an implicit GEMM with 2*B*T2*F2*C*9C FLOPs) from HIP
events recorded around it on the launch stream inside the timed steps.  `cpu_baseline` times the
CPU oracle (the fp32 restatement of the reference encoder) on this host's cores on a bounded
sample (batch 4 x 10 s, the reference's own CPU-sized case).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import conformer_amd as fc                                   # noqa: E402
from conformer_amd import _native as N                       # noqa: E402
from conformer_amd import parallel                           # noqa: E402
from conformer_amd.synth import make_state_dict, synthetic_lengths, synthetic_mel   # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA
FRAME_S = 0.01                 # 10 ms mel hop


def encoder_flops(cfg, B, T):
    """Algorithmic FLOPs of one forward (2 x MACs; the reference executes full T' x T' attention)."""
    m = cfg["model"]
    d, L, C, F = m["d_encoder"], m["encoder_layers"], m["subsampling_channels"], m["d_input"]
    c = lambda n: (n + 1) // 2
    T1, T2, T3 = c(T), c(c(T)), c(c(c(T)))
    F1, F2, F3 = c(F), c(c(F)), c(c(c(F)))
    sub = B * (T1 * F1 * C * 9 + T2 * F2 * C * C * 9 + T3 * F3 * C * C * 9 + T3 * (C * F3) * d)
    layer = B * T3 * d * d * 23 + (2 * T3 - 1) * d * d + B * d * (2 * T3 * T3 + T3 * (2 * T3 - 1)) + 9 * B * T3 * d
    conv2 = B * T2 * F2 * C * C * 9
    return 2 * (sub + L * layer), 2 * conv2


def measured_conv2_traffic(args):
    """HBM bytes per conv2 launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE collected by
    rocprofv3 on this same command, corrected as profiles/r01_conv2_pmc.json records); None for any
    other workload than the one they were taken on."""
    path = os.path.join(ROOT, "profiles", "r01_conv2_pmc.json")
    if args.dtype != "bf16" or args.batch != 64 or args.frames != 2000 or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)["hbm_bytes_per_launch"]


def cpu_baseline(cfg, seed):
    """CPU oracle on a bounded sample: small config, B=4 x T=1000 (10 s), ragged lengths."""
    from oracle import fc_oracle as O
    sd = make_state_dict(cfg, seed)
    lens = [1000, 873, 640, 411]
    x, ln = synthetic_mel(4, 1000, lens, seed + 1)
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    O.encoder_forward(sd, cfg, x, ln)                       # warm-up
    times = []
    t_end = time.perf_counter() + 20.0
    while len(times) < 5 and (time.perf_counter() < t_end or not times):
        t0 = time.perf_counter()
        O.encoder_forward(sd, cfg, x, ln)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(sum(lens) * FRAME_S / med, 1), "unit": "audio-sec/sec", "cores": threads, "kind": "port",
            "sample": f"CPU oracle fp32, fastconformer_small {cfg['model']['encoder_layers']}L, batch 4 x 10 s "
                      f"(lengths {lens}), median of {len(times)} runs, {med * 1e3:.0f} ms/forward",
            "padded_value": round(4 * 1000 * FRAME_S / med, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "fastconformer_small.json"))
    ap.add_argument("--encoder-layers", type=int, default=0, help="override depth (shipped JSON: 17)")
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=2000, help="padded mel frames per utterance (20 s)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=0, help="slices of the batch whose blocks run on concurrent HIP streams (0 = library default)")
    args = ap.parse_args()

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU execution path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = fc.load_config(args.config)
    if args.encoder_layers:
        cfg["model"]["encoder_layers"] = args.encoder_layers
    model = fc.create_model(cfg)
    model.load_state_dict(make_state_dict(cfg, 1234))       # same replica on every rank
    model.encoder.set_precision(args.dtype)
    model = model.to(dev).eval()

    B, T = args.batch, args.frames
    default_workload = (cfg.get("name") == "fastconformer_small" and (B, T, args.dtype) == (64, 2000, "bf16")
                        and not args.encoder_layers)
    lens = synthetic_lengths(B, T, 100 + rank)               # max == T, rest uniform in [T/2, T]
    mel, ln = synthetic_mel(B, T, lens, 200 + rank)
    mel, ln = mel.to(dev), ln.to(dev)
    real_audio_s = sum(lens) * FRAME_S

    pk = model.encoder.packed(dev)
    ws = {}
    stream = torch.cuda.current_stream(dev).cuda_stream
    lib = N.lib()
    events = [(lib.fcenc_event_create(), lib.fcenc_event_create()) for _ in range(args.steps)]

    def forward(ev=None):
        out, out_len, _ = pk.forward(mel, ln, ws, events=ev, lanes=args.lanes)
        return out, out_len

    def run(n, evs=None):
        """n steps.  With several ranks the all-gather of step i runs on RCCL's stream beside the forward of
        step i + 1 (one gather in flight, each forward writes a fresh output tensor); all of them have
        completed when the caller's synchronize returns."""
        pending, res = None, None
        for i in range(n):
            res = forward(evs[i] if evs else None)
            if world > 1:
                started = parallel.gather_outputs_async(res[0], res[1], B * world)
                if pending is not None:
                    pending.wait()
                pending = started
        if pending is not None:
            res = pending.wait()
        return res

    with torch.inference_mode():
        run(args.warmup)
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out, out_len = run(args.steps, events)
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
    assert out.size(0) == B * world and out_len.size(0) == B * world
    assert torch.isfinite(out).all(), "non-finite encoder output"

    conv2_ms = [lib.fcenc_event_elapsed_ms(a, b) for a, b in events]
    for a, b in events:
        lib.fcenc_event_destroy(a)
        lib.fcenc_event_destroy(b)
    stats = torch.tensor([elapsed, real_audio_s, sum(conv2_ms) / len(conv2_ms)], dtype=torch.float64, device=dev)
    if world > 1:
        gathered = [torch.zeros_like(stats) for _ in range(world)]
        torch.distributed.all_gather(gathered, stats)
        elapsed = max(float(g[0]) for g in gathered)
        total_audio = sum(float(g[1]) for g in gathered)
    else:
        total_audio = real_audio_s
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    total_flops, conv2_flops = encoder_flops(cfg, B, T)
    conv2_avg_ms = sum(conv2_ms) / len(conv2_ms)
    achieved = conv2_flops / (conv2_avg_ms * 1e-3) / 1e12
    line = {
        "metric": "encoder audio-sec/sec (RTFx)", "value": round(total_audio / (ms_per_step * 1e-3), 1),
        "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{cfg.get('name', os.path.basename(args.config))} encoder forward, {args.dtype}, "
                               f"batch {B} x {T * FRAME_S:.0f} s per GPU, variable lengths with padding masks"
                               + (" (BASELINE configs[1])" if default_workload else ""),
                   "encoder_layers": cfg["model"]["encoder_layers"], "d_encoder": cfg["model"]["d_encoder"],
                   "batch_per_gpu": B, "global_batch": B * world, "padded_frames": T,
                   "padded_audio_sec_per_sec": round(B * world * T * FRAME_S / (ms_per_step * 1e-3), 1),
                   "gflop_per_forward_per_gpu": round(total_flops / 1e9, 1),
                   "whole_forward_tflops_per_gpu": round(total_flops / (ms_per_step * 1e-3) / 1e12, 1),
                   "whole_forward_mfma_frac": round(total_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                   "parallelism": f"utterance-sharded x{world}, all-gather of outputs+lengths overlapped with the next forward" if world > 1 else "single GPU"},
        "roofline": {"kernel": "conv3x3s2 implicit GEMM (subsampling conv2)", "bound": "mfma",
                     "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": measured_conv2_traffic(args),
                     "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/r01_conv2_pmc.json)",
                     "flops_per_launch": conv2_flops, "avg_launch_ms": round(conv2_avg_ms, 4)},
    }
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(cfg, 1234)
    print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
