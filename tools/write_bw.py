"""Write bandwidth of plain device fills at conv1's output size (1.31 GB), for comparison with conv1's store stream."""
import torch
dev = torch.device("cuda:0")
for name, n, dt in (("bf16 1.31 GB", 64 * 1000 * 40 * 256, torch.bfloat16), ("fp32 1.31 GB", 64 * 1000 * 40 * 128, torch.float32),
                    ("fp32 0.33 GB", 64 * 1000 * 40 * 32, torch.float32)):
    t = torch.empty(n, dtype=dt, device=dev)
    for _ in range(3): t.fill_(1.0)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): t.fill_(2.0)
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / 10 * 1e3
    print(f"fill {name}: {us:.1f} us = {t.numel() * t.element_size() / us / 1e6:.2f} TB/s")
    src = torch.empty_like(t)
    for _ in range(3): t.copy_(src)
    torch.cuda.synchronize()
    a.record()
    for _ in range(10): t.copy_(src)
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / 10 * 1e3
    print(f"copy {name}: {us:.1f} us = {2 * t.numel() * t.element_size() / us / 1e6:.2f} TB/s read+write")
