"""Utterance sharding + all-gather with two gloo processes on CPU.  The encoder itself cannot run on
CPU (HIP only), so the per-rank encode is the autograd composite of the same module under no_grad;
what is under test is the sharding / gather logic: the 2-rank result must equal the 1-rank result
bit for bit, in rank order, for even and ragged shard sizes."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


class _CompositeModel:
    """Stands in for model.encode on CPU ranks."""

    def __init__(self, model):
        self.m = model
        self.d_encoder = model.d_encoder

    def encode(self, x, lengths):
        with torch.no_grad():
            return self.m.encoder.forward_composite(x, lengths)


def _worker(rank, world, port, batch, q):
    sys.path.insert(0, ROOT)
    import conformer_amd as fc
    from conformer_amd import parallel
    from conformer_amd.synth import make_state_dict, synthetic_mel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    meta, _ = load_golden("mini_full")
    m = fc.create_model(meta["config"])
    m.load_state_dict(make_state_dict(meta["config"], 5))
    m.eval()
    lens = [160, 120, 96, 133, 160][:batch]
    x, ln = synthetic_mel(batch, 160, lens, 6)
    out, out_len = parallel.encode_sharded(_CompositeModel(m), x, ln)
    # the benchmark's pipelining: the gather of step i is waited for after step i + 1 has been started
    pipelined_ok, pending = True, None
    for i in range(3):
        mine = torch.full((2, 3, 4), float(10 * i + rank))
        started = parallel.gather_outputs_async(mine, torch.full((2,), 10 * i + rank, dtype=torch.int64), 4)
        if pending is not None:
            got, got_len = pending[1].wait()
            j = pending[0]
            pipelined_ok &= got[:2].eq(10 * j).all().item() and got[2:].eq(10 * j + 1).all().item()
            pipelined_ok &= got_len.tolist() == [10 * j, 10 * j, 10 * j + 1, 10 * j + 1]
        pending = (i, started)
    got, got_len = pending[1].wait()
    pipelined_ok &= got[:2].eq(20).all().item() and got[2:].eq(21).all().item() and got_len.tolist() == [20, 20, 21, 21]
    if rank == 0:
        full, full_len, _ = _CompositeModel(m).encode(x, ln)
        q.put((torch.equal(out, full) and bool(pipelined_ok), torch.equal(out_len, full_len), tuple(out.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [4, 5, 1])
def test_two_rank_sharded_encode_equals_single_rank(batch):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    same_out, same_len, shape = q.get(timeout=5)
    assert same_out and same_len and shape[0] == batch


def test_shard_bounds_cover_batch():
    from conformer_amd.parallel import shard_bounds
    for batch in (1, 7, 64, 65):
        for world in (1, 2, 4, 8):
            spans = [shard_bounds(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
