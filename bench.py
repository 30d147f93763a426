#!/usr/bin/env python3
"""Headline benchmark: FastConformer-small encoder forward, bf16, batch 64 x 20 s per GPU.

    python bench.py [--gpus N --steps K --warmup W]            (N > 1: launched by torch.distributed.run)
    python bench.py --scaling strong --gpus N                  one global batch of --batch utterances cut over the ranks
    python bench.py --workload stream                          cached streaming step, 1024 streams x 160 ms chunks

One step = one encoder forward (`model.encoder(mel, lengths)`: one torch.library operator, one C call into
libfcenc_hip.so) over a batch of synthetic utterances already resident in HBM, followed for N > 1 by the RCCL
all-gather of outputs and lengths (started asynchronously: it overlaps the next step's forward).  Prints ONE JSON
line on rank 0.  `value` = real (unpadded) audio seconds encoded per wall second by the whole job; the padded
figure is in `config`.

`roofline` prices the kernel class with the largest share of the step -- the launch that follows each attention
core (out_proj + conv module + second feed-forward + closing LayerNorm + the next block's first feed-forward and
q/k/v projection: 16 launches, about 40 % of the step) -- from HIP events recorded around the middle block's launch
on the launch stream inside every fourth timed step (each recorded event costs the stream a bubble of about 6 us);
`roofline.kernels` lists the other classes (conv2, the largest single
launch, is timed the same way; the rest come from the committed rocprofv3 summary), and `roofline.sum` is the whole
pass: sum over kernels of max(FLOPs / MFMA peak, bytes / HBM peak) divided by the measured step.  `cpu_baseline`
times the CPU oracle (the fp32 restatement of the reference encoder) on this host's cores on a bounded sample
(batch 4 x 10 s, the reference's own CPU-sized case).
"""
import argparse
import copy
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
PEAK_HBM_GBS = 8000.0          # MI355X HBM3E
FRAME_S = 0.01                 # 10 ms mel hop
PMC_FILE = os.path.join(ROOT, "profiles", "r03_bench_small_bf16_pmc.json")


def c2(n):
    return (n + 1) // 2


def kernel_work(cfg, B, T, es=2):
    """Algorithmic work per kernel class of one forward: {name: (launches, FLOPs per launch, HBM bytes per launch)}.
    FLOPs = 2 x MACs with the reference's full T' x T' attention; bytes = every operand and result once (weights
    once per launch), activations in the compute dtype (es bytes), the residual stream in fp32."""
    m = cfg["model"]
    d, L, C, F, H = m["d_encoder"], m["encoder_layers"], m["subsampling_channels"], m["d_input"], m["encoder_heads"]
    ff = d * m.get("encoder_ff_expansion", 4)
    T1, T2, T3 = c2(T), c2(c2(T)), c2(c2(c2(T)))
    F1, F2, F3 = c2(F), c2(c2(F)), c2(c2(c2(F)))
    M = B * T3
    HD = H * 64
    w = {}
    w["input statistics"] = (1, 0, B * T * F * 4)
    w["conv1 (+ standardise, BN, SiLU)"] = (1, 2 * B * T1 * F1 * C * 9, B * T * F * 4 + B * T1 * F1 * C * es)
    w["conv2 (3x3 s2 implicit GEMM)"] = (1, 2 * B * T2 * F2 * C * C * 9,
                                         (B * T1 * F1 * C + B * T2 * F2 * C + 9 * C * C) * es)
    # the two as one kernel (bf16, 80 mel bins, 256 channels): conv1's activation is neither written nor read
    w["conv1 + conv2 (one kernel)"] = (1, 2 * B * T1 * F1 * C * 9 + 2 * B * T2 * F2 * C * C * 9,
                                       B * T * F * 4 + (B * T2 * F2 * C + 9 * C * C) * es)
    w["conv3 (3x3 s2, patch in LDS)"] = (1, 2 * B * T3 * F3 * C * C * 9,
                                         (B * T2 * F2 * C + B * T3 * F3 * C + 9 * C * C) * es)
    w["subsampling Linear"] = (1, 2 * M * C * F3 * d, (M * C * F3 + C * F3 * d) * es + M * d * 4)
    w["attention core"] = (L, 2 * B * d * (2 * T3 * T3 + T3 * (2 * T3 - 1)),
                           (M * 3 * HD + M * HD + (2 * T3 - 1) * HD) * es)
    # everything of a block that is row-local: two feed-forward modules, q/k/v and out projections, the conv module
    block_w = (2 * 2 * d * ff + 3 * HD * d + d * HD + 2 * d * d + d * d) * es
    w["block (feed-forwards, projections, conv module, LayerNorms)"] = (
        L, 2 * (M * d * d * 23 + 9 * M * d), 2 * M * d * 4 + (M * HD + M * 3 * HD) * es + block_w)
    w["pos_proj of the position table (cached per T')"] = (0, 2 * (2 * T3 - 1) * d * d, 0)
    return w


FUSED12 = "conv1 + conv2 (one kernel)"
SPLIT12 = ("conv1 (+ standardise, BN, SiLU)", "conv2 (3x3 s2 implicit GEMM)")


def scheduled_work(cfg, B, T, fused12):
    """kernel_work without the entries of the schedule that is not run (conv1 and conv2 as one kernel or as two)."""
    w = kernel_work(cfg, B, T)
    for k in ((SPLIT12) if fused12 else (FUSED12,)):
        del w[k]
    return w


def encoder_flops(cfg, B, T):
    """Algorithmic FLOPs of one forward (the same whichever way conv1 and conv2 are scheduled)."""
    w = scheduled_work(cfg, B, T, False)
    L, T3, d = cfg["model"]["encoder_layers"], c2(c2(c2(T))), cfg["model"]["d_encoder"]
    return sum(n * f for n, f, _ in w.values()) + L * 2 * (2 * T3 - 1) * d * d


def roofline_time_ms(cfg, B, T, fused12):
    """sum over kernels of max(FLOPs / MFMA peak, bytes / HBM peak), in ms."""
    return sum(n * max(f / (PEAK_BF16_TFLOPS * 1e12), b / (PEAK_HBM_GBS * 1e9))
               for n, f, b in scheduled_work(cfg, B, T, fused12).values()) * 1e3


def committed_pmc(args):
    """Per-launch HBM bytes by kernel from the committed rocprofv3 counter passes (FETCH_SIZE x 2 and WRITE_SIZE,
    separate passes over this same command); only for the workload they were taken on."""
    if args.dtype != "bf16" or args.batch != 64 or args.frames != 2000 or args.workload != "offline" or not os.path.exists(PMC_FILE):
        return None
    with open(PMC_FILE) as f:
        return json.load(f)


def cpu_threads():
    """Cores this process may use: the job's CPU share when the launcher states one, else the physical cores."""
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or (os.cpu_count() or 1)
    except ImportError:
        physical = os.cpu_count() or 1
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else physical
    return max(1, min(physical, affinity, share or physical)), physical, share


def time_oracle(fn, budget_s, max_runs):
    fn()                                                    # warm-up
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < max_runs and (time.perf_counter() < t_end or not times):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return sorted(times)[len(times) // 2], len(times)


def cpu_baseline(cfg, seed, T=2000):
    """CPU oracle (the fp32 restatement of the reference encoder, pinned to the reference by tests/golden) on bounded
    samples of the workload: the reference's own CPU-sized case (batch 4 x 10 s, BASELINE configs[0] shape) and a
    scaled-down batch of the GPU workload's own shape (8 x 20 s, the first 8 utterances of the benchmark batch)."""
    from oracle import fc_oracle as O
    sd = make_state_dict(cfg, seed)
    threads, physical, share = cpu_threads()
    torch.set_num_threads(threads)
    lens = [1000, 873, 640, 411]
    x, ln = synthetic_mel(4, 1000, lens, seed + 1)
    med, runs = time_oracle(lambda: O.encoder_forward(sd, cfg, x, ln), 12.0, 5)
    lens8 = synthetic_lengths(64, T, 100)[:8]
    x8, ln8 = synthetic_mel(8, T, lens8, seed + 2)
    med8, runs8 = time_oracle(lambda: O.encoder_forward(sd, cfg, x8, ln8), 12.0, 3)
    return {"value": round(sum(lens) * FRAME_S / med, 1), "unit": "audio-sec/sec", "cores": threads, "kind": "port",
            "host_physical_cores": physical, "cpu_share_of_this_job": share or None,
            "sample": f"CPU oracle fp32, fastconformer_small {cfg['model']['encoder_layers']}L, batch 4 x 10 s "
                      f"(lengths {lens}), median of {runs} runs, {med * 1e3:.0f} ms/forward, {threads} threads",
            "padded_value": round(4 * 1000 * FRAME_S / med, 1),
            "gpu_workload_shape": {"value": round(sum(lens8) * FRAME_S / med8, 1), "unit": "audio-sec/sec",
                                   "padded_value": round(8 * T * FRAME_S / med8, 1), "cores": threads,
                                   "sample": f"same oracle, batch 8 x {T * FRAME_S:.0f} s (the first 8 utterances' lengths of the "
                                             f"benchmark batch: {lens8}), median of {runs8} runs, {med8 * 1e3:.0f} ms/forward"}}


def bf16_error_spot_check(model, cfg, mel, ln, out, which=(0, 37)):
    """Relative L2 and worst-element error (over the valid frames) of the timed batch's own output against the CPU
    oracle, for a few utterances (utterances are independent in eval mode, so each is checked on its own)."""
    from oracle import fc_oracle as O
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    res = []
    for b in which:
        if b >= mel.size(0):
            continue
        ref, ref_len, _ = O.encoder_forward(sd, cfg, mel[b:b + 1].cpu(), ln[b:b + 1].cpu())
        n = int(ref_len[0])
        got = out[b, :n].float().cpu()
        err = (got - ref[0, :n]).abs().max().item()
        res.append({"utterance": b, "frames": n, "rel_l2": round(((got - ref[0, :n]).norm() / ref[0, :n].norm()).item(), 5),
                    "worst_element_over_max": round(err / ref[0, :n].abs().max().item(), 5)})
    return res


def build_model(cfg, dtype, dev):
    model = fc.create_model(cfg)
    model.load_state_dict(make_state_dict(cfg, 1234))       # same replica on every rank
    model.encoder.set_precision(dtype)
    return model.to(dev).eval()


def time_forward(model, mel, ln, steps, warmup, dev):
    """ms per forward (median of per-step HIP-event times) for a side measurement outside the timed region."""
    with torch.inference_mode():
        for _ in range(warmup):
            model.encoder(mel, ln)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in evs:
            a.record()
            model.encoder(mel, ln)
            b.record()
        torch.cuda.synchronize(dev)
    return sorted(a.elapsed_time(b) for a, b in evs)[steps // 2]


def bench_stream(args, cfg, dev):
    """cfg5: 1024 concurrent streams, chunks of 17 mel frames (160 ms + look-ahead), caches saturated first."""
    cfg = copy.deepcopy(cfg)
    model = build_model(cfg, args.dtype, dev)
    streams, frames = args.streams, 17
    x, _ = synthetic_mel(streams, frames, None, 2)
    x = x.to(dev)
    cache = None
    lc = cfg["model"].get("local_context", 256)
    saturate = -(-lc // 3) + 4
    with torch.inference_mode():
        for _ in range(saturate + args.warmup):
            _, _, cache = model.encode(x, None, cache=cache, streaming=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out, _, cache = model.encode(x, None, cache=cache, streaming=True)
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    ms = elapsed / args.steps * 1e3
    audio = streams * 0.16                                   # each chunk advances every stream by 160 ms
    m = cfg["model"]
    d, L, C, F, H = m["d_encoder"], m["encoder_layers"], m["subsampling_channels"], m["d_input"], m["encoder_heads"]
    ff, es = d * m.get("encoder_ff_expansion", 4), 2 if args.dtype == "bf16" else 4
    t1, t2, t3 = c2(frames), c2(c2(frames)), c2(c2(c2(frames)))
    f1, f2, f3 = c2(F), c2(c2(F)), c2(c2(c2(F)))
    tok = streams * t3
    # ---- dominant kernel: the attention over the cached keys (one launch per layer) reads both rings once ----
    from conformer_amd import ops
    pk = model.encoder.packed(dev)
    ring_bytes = streams * H * 2 * lc * 64 * es               # K ring + V ring of one layer
    attn_us = None
    if args.dtype == "bf16":
        g = torch.Generator(device=dev).manual_seed(0)
        qkv = (torch.randn((tok, 3 * H * 64), device=dev, generator=g) * 0.5).to(pk.dtype)
        pos = pk.stream_pos_tables(t3)
        n_it = 3
        with torch.inference_mode():
            for l in range(L):                               # warm-up (and the launch is the product's own kernel on the
                ops.stream_attention_(qkv, pos[l], pk.layers[l]["bias_u"], pk.layers[l]["bias_v"], cache.k[l], cache.v[l],
                                      10 * lc, 0.125)        # product's own rings: 17 layers x 268 MB, more than the 256 MB cache)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for it in range(n_it):
                for l in range(L):
                    ops.stream_attention_(qkv, pos[l], pk.layers[l]["bias_u"], pk.layers[l]["bias_v"], cache.k[l], cache.v[l],
                                          10 * lc + t3 * (it + 1), 0.125)
            e1.record()
            torch.cuda.synchronize(dev)
        attn_us = e0.elapsed_time(e1) * 1e3 / (n_it * L)
    # ---- the whole chunk: sum over kernels of max(FLOPs / MFMA peak, bytes / HBM peak) ----
    sub_flops = 2 * streams * (t1 * f1 * C * 9 + t2 * f2 * C * C * 9 + t3 * f3 * C * C * 9 + t3 * f3 * C * d)
    sub_bytes = streams * frames * F * 4 + (streams * (t1 * f1 + t2 * f2 + t3 * f3) * C * 2 + 2 * 9 * C * C + f3 * C * d) * es
    layer_gemm_flops = 2 * (tok * d * d * 23 + 9 * tok * d)
    layer_w_bytes = (2 * 2 * d * ff + 3 * H * 64 * d + d * H * 64 + 3 * d * d) * es
    attn_flops = 2 * streams * H * 64 * t3 * (2 * lc + 2 * t3 - 1)
    roof_ms = (max(sub_flops / (PEAK_BF16_TFLOPS * 1e12), sub_bytes / (PEAK_HBM_GBS * 1e9))
               + L * max(layer_gemm_flops / (PEAK_BF16_TFLOPS * 1e12), layer_w_bytes / (PEAK_HBM_GBS * 1e9))
               + L * max(attn_flops / (PEAK_BF16_TFLOPS * 1e12), ring_bytes / (PEAK_HBM_GBS * 1e9))) * 1e3
    roofline = {"kernel": "stream_attn_fifo_kernel: attention of the chunk's 3 frames over the cached keys, one launch per layer",
                "launches_per_step": L, "bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "algorithmic_bytes_per_launch": ring_bytes,
                "achieved": round(ring_bytes / attn_us * 1e-3, 1) if attn_us else None,
                "frac": round(ring_bytes / attn_us * 1e-3 / PEAK_HBM_GBS, 4) if attn_us else None,
                "avg_launch_ms": round(attn_us / 1e3, 4) if attn_us else None,
                "share_of_step": round(L * attn_us / 1e3 / ms, 3) if attn_us else None,
                "timing": "HIP events around 3 x 17 launches of the kernel on the model's own saturated rings, after the timed steps",
                "traffic": 273.6e6 + 8.7e6 if (streams, lc, args.dtype) == (1024, 256, "bf16") else None,
                "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, profiles/r01_stream_attn_pmc.json)",
                "sum": {"roofline_ms": round(roof_ms, 4), "measured_ms": round(ms, 4), "frac": round(roof_ms / ms, 4),
                        "definition": "sum over kernels of max(FLOPs / 2.5 PFLOP/s, algorithmic HBM bytes / 8 TB/s) / ms_per_step; "
                                      "per layer: the rings once, the block's weights once; the chunk is HBM-bound throughout"}}
    line = {"metric": "encoder audio-sec/sec (RTFx)", "value": round(audio / (ms * 1e-3), 1), "unit": "audio-sec/sec",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{cfg.get('name', 'fastconformer_small')} cached streaming encoder step, {args.dtype}, "
                                   f"{streams} concurrent streams x 160 ms chunks (17 mel frames -> 3 encoder frames), "
                                   f"caches saturated at local_context {lc} (BASELINE configs[4])",
                       "encoder_layers": cfg["model"]["encoder_layers"], "streams": streams,
                       "chunk_latency_ms": round(ms, 4), "fraction_of_real_time": round(ms / 160.0, 5)},
            "roofline": roofline}
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_stream(cfg, 1234, frames, lc)
    print(json.dumps(line), flush=True)


def cpu_baseline_stream(cfg, seed, frames, lc, streams=8):
    """The CPU oracle's cached streaming step (the reference's cat / trim caches, model.py:352-366) for a few streams,
    caches saturated first."""
    from oracle import fc_oracle as O
    sd = make_state_dict(cfg, seed)
    threads, physical, share = cpu_threads()
    torch.set_num_threads(threads)
    x, _ = synthetic_mel(streams, frames, None, seed + 3)
    cache = None
    for _ in range(-(-lc // 3) + 2):
        _, _, cache = O.encoder_forward(sd, cfg, x, None, cache=cache, streaming=True)
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        _, _, cache = O.encoder_forward(sd, cfg, x, None, cache=cache, streaming=True)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(streams * 0.16 / med, 1), "unit": "audio-sec/sec", "cores": threads, "kind": "port",
            "host_physical_cores": physical,
            "sample": f"CPU oracle fp32 streaming step, {streams} streams x 160 ms chunks, caches saturated at {lc} frames, "
                      f"median of 5 chunks, {med * 1e3:.0f} ms per chunk, {threads} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "fastconformer_small.json"))
    ap.add_argument("--encoder-layers", type=int, default=0, help="override depth (shipped JSON: 17)")
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU (weak) or in the global batch (strong)")
    ap.add_argument("--frames", type=int, default=2000, help="padded mel frames per utterance (20 s)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch utterances on every GPU; strong: ONE global batch of --batch utterances, padded "
                         "once, cut over the ranks (the 8-GPU target: 64 utterances -> 8 per GPU)")
    ap.add_argument("--workload", default="offline", choices=["offline", "stream"])
    ap.add_argument("--streams", type=int, default=1024, help="--workload stream: concurrent streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (shard sizes, 12-layer depth)")
    ap.add_argument("--lanes", type=int, default=0, help="slices of the batch whose blocks run on concurrent HIP streams (0 = library default)")
    args = ap.parse_args()

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU execution path)")
    local %= torch.cuda.device_count()       # more ranks than GPUs only in a gloo rehearsal (FCENC_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = fc.load_config(args.config)
    if args.encoder_layers:
        cfg["model"]["encoder_layers"] = args.encoder_layers
    if args.workload == "stream":
        if world > 1:
            raise SystemExit("--workload stream runs on one GPU (streams are independent: replicas only)")
        return bench_stream(args, cfg, dev)
    model = build_model(cfg, args.dtype, dev)
    model.encoder.forward_lanes = args.lanes

    T = args.frames
    if args.scaling == "strong":
        # the global batch is generated and padded ONCE (same seed on every rank), then cut: shards are never re-padded
        g_lens = synthetic_lengths(args.batch, T, 100)
        g_mel, g_ln = synthetic_mel(args.batch, T, g_lens, 200)
        mel, ln = parallel.shard_batch(g_mel, g_ln, world, rank)
        lo, hi = parallel.shard_bounds(args.batch, world, rank)
        lens, global_batch = g_lens[lo:hi], args.batch
        if hi == lo:
            raise SystemExit("--scaling strong: more ranks than utterances")
    else:
        lens = synthetic_lengths(args.batch, T, 100 + rank)      # max == T, rest uniform in [T/2, T]
        mel, ln = synthetic_mel(args.batch, T, lens, 200 + rank)
        global_batch = args.batch * world
    B = mel.size(0)
    mel, ln = mel.contiguous().to(dev), ln.contiguous().to(dev)
    real_audio_s = sum(lens) * FRAME_S
    default_workload = (cfg.get("name") == "fastconformer_small" and (args.batch, T, args.dtype) == (64, 2000, "bf16")
                        and not args.encoder_layers)

    pk = model.encoder.packed(dev)
    lib = N.lib()
    # Every event recorded between two launches costs the stream a bubble of about 6 us.  The kernel-level pairs
    # (conv2, the middle block launch) are therefore recorded on every fourth timed step only, and the step times
    # come from one event per step boundary (N + 1 events), not a pair per step.
    probe_every = 4
    events = [tuple(lib.fcenc_event_create() for _ in range(4)) if i % probe_every == 0 else None
              for i in range(args.steps)]
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]

    def run(n, timed):
        """n steps.  With several ranks the all-gather of step i runs on RCCL's stream beside the forward of
        step i + 1 (one gather in flight, each forward writes a fresh output tensor); all of them have
        completed when the caller's synchronize returns."""
        pending, res = None, None
        for i in range(n):
            pk.profile_events = events[i] if timed else None
            if timed:
                step_ev[i].record()
            out, out_len, _ = model.encoder(mel, ln)            # the product path: fcenc::encoder_forward
            if timed and i == n - 1:
                step_ev[n].record()
            res = (out, out_len)
            if world > 1:
                started = parallel.gather_outputs_async(out, out_len, global_batch)
                if pending is not None:
                    pending.wait()
                pending = started
        pk.profile_events = None
        if pending is not None:
            res = pending.wait()
        return res

    with torch.inference_mode():
        run(args.warmup, False)
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out, out_len = run(args.steps, True)
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
    assert out.size(0) == global_batch and out_len.size(0) == global_batch
    assert torch.isfinite(out).all(), "non-finite encoder output"

    conv2_ms = [lib.fcenc_event_elapsed_ms(e[0], e[1]) for e in events if e is not None]
    block_ms = [lib.fcenc_event_elapsed_ms(e[2], e[3]) for e in events if e is not None]
    for e in events:
        for h in e or ():
            lib.fcenc_event_destroy(h)
    fwd_ms = sorted(step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps))
    stats = torch.tensor([elapsed, real_audio_s], dtype=torch.float64, device=dev)
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
    total_flops = encoder_flops(cfg, B, T)
    fused12 = bool(pk.conv12 and pk.fuse_conv12)          # conv1 produced inside conv2: the timed launch is that one kernel
    work = scheduled_work(cfg, B, T, fused12)
    L = cfg["model"]["encoder_layers"]
    conv2_avg_ms = sum(conv2_ms) / len(conv2_ms)
    block_avg_ms = sum(block_ms) / len(block_ms)
    # the timed launch also carries the NEXT block's first feed-forward and q/k/v projection and not its own: the
    # same work as one whole block's row-local part (the last block's launch has no successor; the first block's
    # feed-forward is its own launch)
    _, block_flops, block_bytes = work["block (feed-forwards, projections, conv module, LayerNorms)"]
    fused_block = bool(lib.fcenc_convmod_oproj_supported(pk.code, pk.d, 9, pk.heads)
                       and lib.fcenc_ffn_qkv_fused_supported(pk.code, pk.d, pk.ff, pk.heads))
    pmc = committed_pmc(args) or {}
    front_name = FUSED12 if fused12 else SPLIT12[1]
    _, conv2_flops, conv2_bytes = work[front_name]
    kernels = [
        {"kernel": front_name + ", 1 launch", "bound": "mfma", "avg_launch_ms": round(conv2_avg_ms, 4),
         "achieved": round(conv2_flops / (conv2_avg_ms * 1e-3) / 1e12, 1), "unit": "TFLOP/s",
         "frac": round(conv2_flops / (conv2_avg_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
         "share_of_step": round(conv2_avg_ms / ms_per_step, 3), "flops_per_launch": conv2_flops,
         "algorithmic_bytes_per_launch": conv2_bytes,
         "traffic": pmc.get("conv12_hbm_bytes_per_launch" if fused12 else "conv2_hbm_bytes_per_launch")},
    ]
    if fused12:
        kernels[0]["note"] = ("12 x 20 output pixels x 256 channels per workgroup; conv1 (MFMA on a bf16 head + remainder split, SiLU) is "
                              "produced into LDS by the same waves beside conv2's MFMAs, so the kernel reads the mel input and "
                              "conv2's weights only: timing-only builds give 582 us for conv2's MFMAs alone and 210 us more for "
                              "the producer, whose SiLUs share the SIMDs' issue slots with the MFMAs")
    elif pk.channels == 256 and args.dtype == "bf16":
        # conv2 runs 256 x 256 tiles, one workgroup per CU: per 64-wide K slice a workgroup pulls 32 KiB of im2col rows and
        # 32 KiB of weights through its CU's load path (timing-only builds: without those loads the kernel is 38 % shorter,
        # without its LDS fragment reads 8 %), i.e. 2 x 9 x C x 256 x 2 bytes per workgroup
        t2 = c2(c2(T))
        wg2 = -(-B * t2 * c2(c2(pk.n_mels)) // 256)
        wg2_bytes = 2 * 9 * pk.channels * 256 * 2
        floor2_ms = -(-wg2 // 256) * wg2_bytes / 70e9 * 1e3
        kernels[0]["load_path"] = {"bytes_per_workgroup": wg2_bytes, "workgroups": wg2, "per_cu_rate_GBps": 70.0,
                                   "floor_ms": round(floor2_ms, 4), "frac": round(floor2_ms / conv2_avg_ms, 4),
                                   "definition": "rounds x bytes one workgroup stages through its CU's load path / 70 GB/s, "
                                                 "over the measured launch"}
    for name, key, bound in (("conv1 (+ standardise, BN, SiLU)", "conv1", "hbm"), ("input statistics", "stats", "hbm"),
                             ("conv3 (3x3 s2, patch in LDS)", "conv3", "mfma"), ("attention core", "attention", "mfma"),
                             ("subsampling Linear", "linear", "mfma")):
        us = pmc.get(key + "_avg_us")
        if us is None or name not in work:
            continue
        n, fl, by = work[name]
        if bound == "hbm":
            ach = by / (us * 1e-6) / 1e9
            kernels.append({"kernel": f"{name}, {n} launch(es)", "bound": "hbm", "avg_launch_ms": round(us / 1e3, 4),
                            "achieved": round(ach, 1), "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                            "share_of_step": round(n * us / 1e3 / ms_per_step, 3), "traffic": pmc.get(key + "_hbm_bytes_per_launch"),
                            "source": "rocprofv3 (profiles/)"})
        else:
            ach = fl / (us * 1e-6) / 1e12
            kernels.append({"kernel": f"{name}, {n} launch(es)", "bound": "mfma", "avg_launch_ms": round(us / 1e3, 4),
                            "achieved": round(ach, 1), "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                            "share_of_step": round(n * us / 1e3 / ms_per_step, 3), "traffic": pmc.get(key + "_hbm_bytes_per_launch"),
                            "source": "rocprofv3 (profiles/)"})
    rt_ms = roofline_time_ms(cfg, B, T, fused12)
    block_ach = block_flops / (block_avg_ms * 1e-3) / 1e12
    # what actually bounds the fused block launch: every 64-token workgroup pulls the block's weights and its own
    # activations through its CU's L2 port (66-73 GB/s per CU measured for rows served by the L2), one workgroup
    # per CU and ceil(workgroups / 256) rounds
    l2_stream = None
    if fused_block:
        d_enc = cfg["model"]["d_encoder"]
        wg = B * ((c2(c2(c2(T))) + 63) // 64)              # utterances are padded to whole 64-frame tiles
        wg_bytes = kernel_work(cfg, B, T)["block (feed-forwards, projections, conv module, LayerNorms)"][2] - 2 * B * c2(c2(c2(T))) * d_enc * 4 \
            - 4 * B * c2(c2(c2(T))) * d_enc * 2
        wg_bytes += 64 * (d_enc * 2 + d_enc * 4)           # + its rows of the attention output and of the fp32 stream
        floor_ms = -(-wg // 256) * wg_bytes / 70e9 * 1e3
        l2_stream = {"bytes_per_workgroup": wg_bytes, "workgroups": wg, "per_cu_rate_GBps": 70.0, "floor_ms": round(floor_ms, 4),
                     "frac": round(floor_ms / block_avg_ms, 4),
                     "definition": "rounds x bytes one workgroup reads through its CU's L2 port / 70 GB/s, over the measured launch"}
    line = {
        "metric": "encoder audio-sec/sec (RTFx)", "value": round(total_audio / (ms_per_step * 1e-3), 1),
        "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{cfg.get('name', os.path.basename(args.config))} encoder forward, {args.dtype}, "
                               + (f"batch {args.batch} x {T * FRAME_S:.0f} s per GPU" if args.scaling == "weak" else
                                  f"ONE global batch {args.batch} x {T * FRAME_S:.0f} s cut over {world} GPU(s)")
                               + ", variable lengths with padding masks"
                               + (" (BASELINE configs[1])" if default_workload else ""),
                   "encoder_layers": L, "d_encoder": cfg["model"]["d_encoder"],
                   "batch_per_gpu": B, "global_batch": global_batch, "padded_frames": T,
                   "padded_audio_sec_per_sec": round(global_batch * T * FRAME_S / (ms_per_step * 1e-3), 1),
                   "forward_ms_min_median_max": [round(fwd_ms[0], 4), round(fwd_ms[len(fwd_ms) // 2], 4), round(fwd_ms[-1], 4)],
                   "gflop_per_forward_per_gpu": round(total_flops / 1e9, 1),
                   "whole_forward_tflops_per_gpu": round(total_flops / (ms_per_step * 1e-3) / 1e12, 1),
                   "whole_forward_mfma_frac": round(total_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                   "parallelism": (f"utterance-sharded x{world} ({args.scaling} scaling), all-gather of outputs+lengths "
                                   "overlapped with the next forward") if world > 1 else "single GPU"},
        "roofline": {"kernel": ("block launch after each attention core: out_proj + conv module + feed-forward 2 + closing "
                                "LayerNorm + next block's feed-forward 1 + q/k/v projection" if fused_block else
                                "rest of a block after its attention core (unfused launches)"),
                     "launches_per_step": L - 1 if fused_block else L,
                     "share_of_step": round((L - 1 if fused_block else L) * block_avg_ms / ms_per_step, 3),
                     "bound": "mfma", "achieved": round(block_ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(block_ach / PEAK_BF16_TFLOPS, 4), "traffic": pmc.get("block_hbm_bytes_per_launch"),
                     "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes, profiles/r03_bench_small_bf16_pmc.json)",
                     "mfma_busy": pmc.get("block_mfma_busy"),
                     "flops_per_launch": block_flops, "algorithmic_bytes_per_launch": block_bytes,
                     "avg_launch_ms": round(block_avg_ms, 4),
                     "note": "weights (2.9 MB per launch at d = 256) are re-streamed from L2 by every workgroup; that stream, "
                             "not HBM or MFMA, paces the kernel",
                     "l2_stream": l2_stream,
                     "kernels": kernels,
                     "sum": {"roofline_ms": round(rt_ms, 4), "measured_ms": round(ms_per_step, 4),
                             "frac": round(rt_ms / ms_per_step, 4),
                             "definition": "sum over kernels of max(FLOPs / 2.5 PFLOP/s, algorithmic HBM bytes / 8 TB/s) / ms_per_step"}},
    }
    if world == 1 and not args.no_extras and args.scaling == "weak":
        extras = {}
        # what one GPU does at the shard sizes of a global batch cut over 2 / 4 / 8 GPUs (the 8-GPU target: 8 per GPU)
        shard = {}
        for div in (2, 4, 8):
            b = args.batch // div
            if b >= 1:
                shard[str(b)] = round(time_forward(model, mel[:b].contiguous(), ln[:b].contiguous(), 10, 2, dev), 4)
        extras["forward_ms_at_shard_batch"] = shard
        if not args.encoder_layers and L != 12 and cfg.get("name") == "fastconformer_small":
            # the depth BASELINE.json names for this model (the shipped JSON says 17)
            c12 = copy.deepcopy(cfg)
            c12["model"]["encoder_layers"] = 12
            m12 = build_model(c12, args.dtype, dev)
            ms12 = time_forward(m12, mel, ln, 10, 2, dev)
            f12 = encoder_flops(c12, B, T)
            extras["depth_12_layers"] = {"ms_per_forward": round(ms12, 4),
                                         "audio_sec_per_sec": round(real_audio_s / (ms12 * 1e-3), 1),
                                         "gflop_per_forward": round(f12 / 1e9, 1)}
            del m12
        line["config"].update(extras)
    if not args.no_cpu_baseline:
        if world == 1 and args.scaling == "weak":
            # parity beside the throughput: the timed batch's own output against the oracle (tolerance of the path: 2e-2)
            line["config"][f"{args.dtype}_error_vs_cpu_oracle"] = bf16_error_spot_check(model, cfg, mel, ln, out)
        line["cpu_baseline"] = cpu_baseline(cfg, 1234, T)
    print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
