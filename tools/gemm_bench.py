"""Times the GEMM / conv kernel on the encoder's shapes (cfg2) for the current FCENC_GEMM_VARIANT."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from conformer_amd import ops, _native as N

dev = torch.device("cuda:0")
M = 16384
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us

res = []
dt = torch.bfloat16
for name, n, k, epi in [("ffn1", 1024, 256, N.EPI_STORE_SILU), ("qkv", 768, 256, N.EPI_STORE), ("glu", 512, 256, N.EPI_GLU)]:
    a = torch.randn(M, k, device=dev).to(dt); w = (torch.randn(n, k, device=dev) / k ** 0.5).to(dt); b = torch.randn(n, device=dev)
    us = timeit(lambda: ops.linear(a, w, b, epi))
    res.append((name, us, 2 * M * n * k / us / 1e6))
for name, n, k in [("ffn2", 256, 1024), ("out", 256, 256), ("s4", 256, 2560)]:
    a = torch.randn(M, k, device=dev).to(dt); w = (torch.randn(n, k, device=dev) / k ** 0.5).to(dt); b = torch.randn(n, device=dev)
    x = torch.randn(M, n, device=dev); xc = torch.empty(M, n, device=dev, dtype=dt)
    us = timeit(lambda: ops.linear_residual_(x, a, w, b, 0.5, 1.0, xc))
    res.append((name, us, 2 * M * n * k / us / 1e6))
for name, B, H, W in [("conv2", 64, 1000, 40), ("conv3", 64, 500, 20)]:
    x = torch.randn(B, H, W, 256, device=dev).to(dt); w = (torch.randn(256, 2304, device=dev) / 48).to(dt); b = torch.randn(256, device=dev)
    us = timeit(lambda: ops.conv3x3s2_bn_silu(x, w, b, 0), 10)
    res.append((name, us, 2 * B * ((H + 1) // 2) * ((W + 1) // 2) * 256 * 2304 / us / 1e6))
print("variant", os.environ.get("FCENC_GEMM_VARIANT", "0"), " ".join(f"{n}:{us:.1f}us/{tf:.0f}TF" for n, us, tf in res))
