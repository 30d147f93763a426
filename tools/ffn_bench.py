"""Times the fused feed-forward kernel and the fused convolution module (d = 256)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from conformer_amd import ops
dev = torch.device("cuda:0")
d = 256


def timeit(run, n=50):
    for _ in range(5): run()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): run()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


for M in (16384, 4096):
    x = torch.randn(M, d, device=dev); g = torch.ones(d, device=dev); b = torch.zeros(d, device=dev)
    for ff in (1024, 2048):
        w1 = (torch.randn(ff, d, device=dev) / 16).bfloat16(); w2 = (torch.randn(d, ff, device=dev) / 32).bfloat16()
        w1, w2 = ops.ffn_pack(w1, w2)
        b1 = torch.zeros(ff, device=dev); b2 = torch.zeros(d, device=dev); xc = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
        us = timeit(lambda: ops.ffn_fused_(x, g, b, w1, b1, w2, b2, g, b, 0.5, xc))
        print(f"ffn M={M} ff={ff}: {us:.1f} us  {4*M*d*ff/us/1e6:.0f} TFLOP/s", flush=True)
    w1 = (torch.randn(2 * d, d, device=dev) / 16).bfloat16(); w2 = (torch.randn(d, d, device=dev) / 16).bfloat16()
    w1, w2 = ops.convmod_pack(w1, w2)
    b1 = torch.zeros(2 * d, device=dev); b2 = torch.zeros(d, device=dev)
    dw =torch.randn(9, d, device=dev) * 0.3; dwb = torch.zeros(d, device=dev)
    us = timeit(lambda: ops.convmod_fused(x, g, b, w1, b1, dw, dwb, w2, b2, 256))
    print(f"convmod M={M}: {us:.1f} us", flush=True)
