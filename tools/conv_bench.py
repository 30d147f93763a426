"""Times the 3x3 stride-2 subsampling convolutions (conv2 / conv3 shapes of the small config) and checks
the result against torch's convolution."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from conformer_amd import ops

dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dt = torch.bfloat16
flags = sys.argv[1:] or ["0"]
for name, B, H, W in [("conv2", 64, 1000, 40), ("conv3", 64, 500, 20)]:
    x = torch.randn(B, H, W, 256, device=dev).to(dt); w = (torch.randn(256, 2304, device=dev) / 48).to(dt); b = torch.randn(256, device=dev)
    ref = None
    for f in flags:
        os.environ["FCENC_CONV_DBG"] = f
        us = timeit(lambda: ops.conv3x3s2_bn_silu(x, w, b, 0))
        y = ops.conv3x3s2_bn_silu(x, w, b, 0)
        if ref is None: ref = y
        diff = (y.float() - ref.float()).abs().max().item()
        tf = 2 * B * ((H + 1) // 2) * ((W + 1) // 2) * 256 * 2304 / us / 1e6
        print(f"{name} flags={f}: {us:.1f} us  {tf:.0f} TF/s  maxdiff_vs_first={diff:.3g}", flush=True)
