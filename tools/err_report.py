"""Prints HIP-vs-oracle error of the encoder output per fixture and precision (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, copy
from conftest import load_golden
import conformer_amd as fc
from conformer_amd.synth import make_state_dict, synthetic_mel
from oracle import fc_oracle as O

for name in ["mini_full", "mini_nolen", "small17", "large4", "medium_local"]:
    meta, arr = load_golden(name)
    cfg = meta["config"]
    sd = make_state_dict(cfg, meta["weight_seed"])
    x, ln = synthetic_mel(meta["B"], meta["T"], meta["lengths"], meta["input_seed"])
    L = cfg["model"]["encoder_layers"]
    for nl in sorted(set([1, 2, 4, L])):
        if nl > L: continue
        c2 = copy.deepcopy(cfg); c2["model"]["encoder_layers"] = nl
        ref, _, _ = O.encoder_forward(sd, c2, x, ln)
        for prec in ["fp32", "bf16"]:
            m = fc.create_model(c2); m.load_state_dict(sd, strict=False); m.encoder.set_precision(prec); m = m.cuda().eval()
            with torch.inference_mode():
                out, _, _ = m.encode(x.cuda(), None if ln is None else ln.cuda())
            out = out.cpu()
            e = (out - ref).abs()
            print(f"{name:13s} L={nl:2d} {prec}: max_abs={e.max():.3e} rel_inf={e.max()/ref.abs().max():.3e} "
                  f"rel_l2={(out-ref).norm()/ref.norm():.3e} ref_max={ref.abs().max():.2f} p99.9={e.flatten().kthvalue(int(e.numel()*0.999)).values:.3e}")
