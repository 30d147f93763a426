"""One-rank RCCL check of the asynchronous output gather (the multi-GPU bench path on a single GPU):
    python tools/nccl_gather_check.py
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conformer_amd import parallel          # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
out = torch.randn(8, 250, 256, device="cuda")
ln = torch.arange(8, device="cuda", dtype=torch.int64)
pending = None
for i in range(3):
    started = parallel.gather_outputs_async(out + i, ln + i, 8)
    if pending is not None:
        a, b = pending[1].wait()
        assert torch.equal(a, out + pending[0]) and torch.equal(b, ln + pending[0])
    pending = (i, started)
a, b = pending[1].wait()
torch.cuda.synchronize()
assert torch.equal(a, out + 2) and torch.equal(b, ln + 2)
dist.barrier()
dist.destroy_process_group()
print("rccl async gather ok")
