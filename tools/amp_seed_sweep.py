"""AMP gradient error (relative L2 against ATen fp32) of the HIP training path and of ATen bf16 autocast over several
random initialisations of the small transducer step of tests/test_gpu_train.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conformer_amd as fc  # noqa: E402
from conformer_amd.synth import make_state_dict, synthetic_mel  # noqa: E402

CFG = {"model": {"d_encoder": 128, "encoder_layers": 3, "encoder_heads": 2, "subsampling_channels": 128,
                 "local_context": 16, "dropout": 0.0, "stochastic_depth_prob": 0.0}}


def grads(seed, native, dtype):
    m = fc.create_model(CFG)
    m.load_state_dict(make_state_dict(CFG, seed))
    m = m.cuda().train()
    m.encoder.native_training = native
    x, ln = synthetic_mel(2, 160, [160, 120], seed + 100)
    tokens = torch.tensor([[3, 7, 1, 9], [5, 2, 0, 0]]).cuda()
    tl = torch.tensor([4, 2]).cuda()
    if dtype is None:
        logits, _ = m(x.cuda(), ln.cuda(), tokens, tl)
    else:
        with torch.autocast("cuda", dtype=dtype):
            logits, _ = m(x.cuda(), ln.cuda(), tokens, tl)
    logits.float().log_softmax(-1)[..., 0].sum().backward()
    return {k: p.grad.detach().double() for k, p in m.named_parameters() if p.grad is not None}


def err(g, ref):
    return (sum((g[k] - ref[k]).pow(2).sum().item() for k in ref) / sum(v.pow(2).sum().item() for v in ref.values())) ** 0.5


hip, aten = [], []
for seed in range(40, 52):
    ref = grads(seed, False, None)
    hip.append(err(grads(seed, True, torch.bfloat16), ref))
    aten.append(err(grads(seed, False, torch.bfloat16), ref))
    print(f"seed {seed}: HIP bf16 {hip[-1]:.3e}   ATen bf16 {aten[-1]:.3e}", flush=True)
print(f"median HIP {sorted(hip)[len(hip) // 2]:.3e}  ATen {sorted(aten)[len(aten) // 2]:.3e};  "
      f"HIP better in {sum(h < a for h, a in zip(hip, aten))} of {len(hip)}")
