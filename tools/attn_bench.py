"""Times the relative-position attention kernel on the small config's bench shape (B=64, H=4, T'=250)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from conformer_amd import ops

dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

B, H, Tp, Tv = 64, 4, 256, 250
torch.manual_seed(0)
dt = torch.bfloat16
qkv = torch.randn(B * Tp, 3 * H * 64, device=dev).to(dt)
pos = torch.randn(2 * Tp + 64, H * 64, device=dev).to(dt)
bu = torch.randn(H * 64, device=dev) * 0.1; bv = torch.randn(H * 64, device=dev) * 0.1
klen = torch.randint(Tv // 2, Tv + 1, (B,), device=dev, dtype=torch.int32); klen[0] = Tv
for f in (sys.argv[1:] or ["0"]):
    os.environ["FCENC_ATTN_DBG"] = f
    for window in (0, 128):
        us = timeit(lambda: ops.relpos_attention(qkv, pos, bu, bv, klen, B, Tv, window, 0.125))
        print(f"flags={f} window={window}: {us:.1f} us", flush=True)
