"""Where the block kernel's time goes: run the default bench workload through the library named by FCENC_LIB_PATH
(a variant from tools/diag_build.sh, or the shipped one) and print the block launch's duration; with the `stamps`
variant also the phase clocks of the last launch.

    FCENC_LIB_PATH=tools/diag/libfcenc_stamps.so python tools/diag_block.py
    FCENC_LIB_PATH=tools/diag/libfcenc_noload.so python tools/diag_block.py     # results are wrong, timing is the point
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from conformer_amd import _native as N  # noqa: E402
import conformer_amd as fc  # noqa: E402

SLOTS = {0: "start", 1: "O tile in LDS", 2: "out_proj MFMAs", 3: "projection in LDS", 4: "x+=, LayerNorm", 5: "GEMM-1",
         6: "GLU", 7: "depthwise+SiLU", 8: "GEMM-2", 9: "conv epilogue"}
for base, name in ((16, "ffn2"), (32, "ffn1'")):
    SLOTS[base] = f"{name} entry"
    SLOTS[base + 1] = f"{name} LayerNorm"
    for c in range(4):
        SLOTS[base + 2 + 2 * c] = f"{name} chunk {c} GEMM-1 (+GEMM-2 of chunk {c - 1})" if c else f"{name} chunk 0 GEMM-1"
        SLOTS[base + 3 + 2 * c] = f"{name} chunk {c} SiLU -> LDS"
    SLOTS[base + 10] = f"{name} last GEMM-2"
    SLOTS[base + 11] = f"{name} epilogue"
    SLOTS[base + 12] = f"{name} q|k|v projection (last block)"
    SLOTS[base + 13] = f"{name} q|k|v projection block 0"
    SLOTS[base + 14] = f"{name} q|k|v projection block 1"
SLOTS[48] = "end"


def main():
    batch = int(os.environ.get("DIAG_BATCH", "64"))
    dev = torch.device("cuda", 0)
    cfg = fc.load_config(os.path.join(os.path.dirname(bench.__file__), "configs", "fastconformer_small.json"))
    model = bench.build_model(cfg, "bf16", dev)
    lens = bench.synthetic_lengths(batch, 2000, 100)
    mel, ln = bench.synthetic_mel(batch, 2000, lens, 200)
    mel, ln = mel.contiguous().to(dev), ln.contiguous().to(dev)
    pk = model.encoder.packed(dev)
    lib = N.lib()
    steps = 20
    events = [tuple(lib.fcenc_event_create() for _ in range(4)) for _ in range(steps)]
    fwd = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    with torch.inference_mode():
        for _ in range(5):
            model.encoder(mel, ln)
        torch.cuda.synchronize()
        for i in range(steps):
            pk.profile_events = events[i]
            fwd[i][0].record()
            model.encoder(mel, ln)
            fwd[i][1].record()
        pk.profile_events = None
        torch.cuda.synchronize()
    blk = sorted(lib.fcenc_event_elapsed_ms(e[2], e[3]) * 1e3 for e in events)
    f = sorted(a.elapsed_time(b) for a, b in fwd)
    out = {"lib": os.path.basename(N.LIB_PATH), "batch": batch, "block_us_min_median_max": [round(blk[0], 1), round(blk[steps // 2], 1), round(blk[-1], 1)],
           "forward_ms_median": round(f[steps // 2], 3)}
    print(json.dumps(out), flush=True)

    raw = C.CDLL(N.LIB_PATH)
    if not hasattr(raw, "fcenc_debug_stamps"):
        return
    nwg = batch * 256 // 64
    buf = np.zeros((nwg, 8, 64), dtype=np.uint64)
    raw.fcenc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
    got = raw.fcenc_debug_stamps(buf.ctypes.data, nwg)
    assert got == nwg, got
    t = buf.astype(np.int64)
    t0 = t[:, :, 0].min()                       # the s_memtime clock is one counter for the chip
    print(f"launch span in clocks: first start {0}, last start {int((t[:, :, 0]).max() - t0)}, "
          f"first end {int(t[:, :, 48].min() - t0)}, last end {int(t[:, :, 48].max() - t0)}")
    order = {44: 46.5}                            # the projection's last stamp follows those of its first two blocks
    slots = [s for s in sorted(SLOTS, key=lambda k: order.get(k, k)) if (buf[:, :, s] != 0).all()]      # a phase the kernel does not have leaves its slot empty
    print(f"{'phase (clocks since the previous stamp)':58s} {'wave0 mean':>10s} {'all mean':>9s} {'min':>7s} {'max':>7s}   cumulative(all mean)")
    prev = t[:, :, 0]
    cum0 = t[:, :, 0]
    for s in slots[1:]:
        cur = t[:, :, s]
        d = cur - prev
        print(f"{s:3d} {SLOTS[s]:54s} {d[:, 0].mean():10.0f} {d.mean():9.0f} {d.min():7d} {d.max():7d}   {(cur - cum0).mean():9.0f}")
        prev = cur
    np.save(os.path.join("gpurun_out", "stamps.npy"), buf)


if __name__ == "__main__":
    main()
