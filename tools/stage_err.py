"""Front-end stage errors (bf16 vs oracle) on the GPU box."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_golden
import conformer_amd as fc
from conformer_amd import ops, _native as N
from conformer_amd.synth import make_state_dict, synthetic_mel
from oracle import fc_oracle as O

def rel(a, b): return ((a.float().cpu() - b).norm() / b.norm()).item()
for name in ["small17", "large4"]:
    meta, arr = load_golden(name); cfg = meta["config"]
    sd = make_state_dict(cfg, meta["weight_seed"])
    x, ln = synthetic_mel(meta["B"], meta["T"], meta["lengths"], meta["input_seed"])
    tr = {}
    O.encoder_forward(sd, cfg, x, ln, trace=tr)
    for prec, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        m = fc.create_model(cfg); m.load_state_dict(sd); m = m.cuda().eval()
        pk = m.encoder.packed(torch.device("cuda:0"), dt)
        c1 = ops.lognorm_conv1(x.cuda(), pk.conv1_w9, pk.conv1_b, dt == torch.bfloat16)
        c2 = ops.conv3x3s2_bn_silu(c1, pk.conv_w[0], pk.conv_b[0], 0)
        c3 = ops.conv3x3s2_bn_silu(c2, pk.conv_w[1], pk.conv_b[1], 0)
        B, T3, F3, C = c3.shape
        emb = ops.linear(c3.view(B * T3, F3 * C), pk.lin_w, pk.lin_b, N.EPI_STORE).float() * (pk.d ** 0.5)
        print(name, prec, "c1", rel(c1.permute(0, 3, 1, 2), tr["conv1"]), "c2", rel(c2.permute(0, 3, 1, 2), tr["conv2"]),
              "c3", rel(c3.permute(0, 3, 1, 2), tr["conv3"]), "embed", rel(emb.view(B, T3, -1), tr["embed"]))
        # exact-input check of each stage alone: feed oracle activations
        c2x = ops.conv3x3s2_bn_silu(tr["conv1"].permute(0, 2, 3, 1).contiguous().to(dt).cuda(), pk.conv_w[0], pk.conv_b[0], 0)
        c3x = ops.conv3x3s2_bn_silu(tr["conv2"].permute(0, 2, 3, 1).contiguous().to(dt).cuda(), pk.conv_w[1], pk.conv_b[1], 0)
        if dt == torch.bfloat16 and pk.conv12 is not None:
            # the shipped front-end: conv1 inside conv2 (one kernel), conv3 with its patch in LDS
            f2 = ops.lognorm_conv12(x.cuda(), pk.conv1_w9, pk.conv1_b, pk.conv_w[0], pk.conv_b[0])
            f3 = ops.conv3_patch(f2, pk.conv_w[1], pk.conv_b[1], 0)
            femb = ops.linear(f3.view(B * T3, F3 * C), pk.lin_w, pk.lin_b, N.EPI_STORE).float() * (pk.d ** 0.5)
            print("   fused front-end: c2", rel(f2.permute(0, 3, 1, 2), tr["conv2"]), "c3", rel(f3.permute(0, 3, 1, 2), tr["conv3"]),
                  "embed", rel(femb.view(B, T3, -1), tr["embed"]))
        print("   stage alone: c2", rel(c2x.permute(0, 3, 1, 2), tr["conv2"]), "c3", rel(c3x.permute(0, 3, 1, 2), tr["conv3"]))
        print("   scales: c1 rms", tr["conv1"].pow(2).mean().sqrt().item(), "c2", tr["conv2"].pow(2).mean().sqrt().item(),
              "c3", tr["conv3"].pow(2).mean().sqrt().item(), "embed", tr["embed"].pow(2).mean().sqrt().item())
