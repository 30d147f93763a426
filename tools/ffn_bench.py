import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from conformer_amd import ops
dev = torch.device("cuda:0")
d = 256
for M in (16384, 4096):
  for ff in (256, 512, 1024, 2048):
    x = torch.randn(M, d, device=dev); g = torch.ones(d, device=dev); b = torch.zeros(d, device=dev)
    w1 = (torch.randn(ff, d, device=dev) / 16).bfloat16(); w2 = (torch.randn(d, ff, device=dev) / 32).bfloat16()
    b1 = torch.zeros(ff, device=dev); b2 = torch.zeros(d, device=dev); xc = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    def run(): ops.ffn_fused_(x, g, b, w1, b1, w2, b2, g, b, 0.5, xc)
    for _ in range(5): run()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): run()
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / 50 * 1e3
    print("dbg", os.environ.get("FCENC_FFN_DBG", "0"), f"M={M} ff={ff}: {us:.1f} us  {4*M*d*ff/us/1e6:.0f} TFLOP/s")
