"""Times conv1 (standardise + 1->256 conv + SiLU, bf16 out) at the bench shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from conformer_amd import ops
from conformer_amd.synth import synthetic_mel, synthetic_lengths
dev = torch.device("cuda:0")
B, T = int(os.environ.get('C1_BATCH', '64')), 2000
mel, _ = synthetic_mel(B, T, synthetic_lengths(B, T, 1), 2)
mel = mel.to(dev)
w = (torch.randn(9, 256, device=dev) * 0.3); b = torch.randn(256, device=dev) * 0.1
def run(): return ops.lognorm_conv1(mel, w, b, True)
for _ in range(3): run()
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
us = a.elapsed_time(e) / 10 * 1e3
print(f"stats+conv1: {us:.1f} us  ({B*1000*40*256*2/us/1e6:.2f} TB/s written)")
