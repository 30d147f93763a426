"""Multi-rank encode on the HIP kernels against the single-rank encode, bit for bit (run under torch.distributed.run).

    FCENC_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29611 tools/sharded_check.py

Every rank encodes its contiguous slice of ONE globally padded batch (`parallel.encode_sharded`) and gathers; rank 0
also encodes the whole batch alone and compares.  With FCENC_DIST_BACKEND=gloo the ranks may share a GPU (the gather
is staged through host memory); on a multi-GPU node the default backend is RCCL and nothing else changes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conformer_amd as fc                                          # noqa: E402
from conformer_amd import parallel                                  # noqa: E402
from conformer_amd.synth import make_state_dict, synthetic_lengths, synthetic_mel   # noqa: E402


def main():
    rank, world, local = parallel.init_from_env()
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    res = {}
    for name, batch, frames in (("fastconformer_small", 11, 640), ("fastconformer_medium", 5, 2100)):
        cfg = fc.load_config(os.path.join(os.path.dirname(__file__), "..", "configs", name + ".json"))
        cfg["model"]["encoder_layers"] = 3
        model = fc.create_model(cfg)
        model.load_state_dict(make_state_dict(cfg, 9))
        model.encoder.set_precision("bf16")
        model = model.to(dev).eval()
        t_out = -(-frames // 8)
        lens = synthetic_lengths(batch, frames, 3, min_frames=8 * max(1, t_out - cfg["model"]["local_context"] + 1))
        mel, ln = synthetic_mel(batch, frames, lens, 4)              # same global batch on every rank
        mel, ln = mel.to(dev), ln.to(dev)
        with torch.inference_mode():
            out, out_len = parallel.encode_sharded(model, mel, ln)
            if rank == 0:
                full, full_len, _ = model.encode(mel, ln)
                res[name] = {"equal": bool(torch.equal(out, full) and torch.equal(out_len, full_len)),
                             "shape": list(out.shape), "world": world}
    torch.distributed.barrier()
    if rank == 0:
        print(json.dumps(res), flush=True)
        assert all(v["equal"] for v in res.values()), res
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
