"""Greedy RNN-T decode on the device (`conformer_amd.decode`, one launch of `fcenc_greedy_decode` per
batch) against (a) the token sequences the reference's own loop produced from the reference's encoder
output (committed fixtures) and (b) the CPU oracle on further shapes.  Tokens are integers: identical,
unless a decision was a near-tie below fp32 summation-order noise (the smallest top-2 gap is checked)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
import conformer_amd as fc
from conformer_amd import decode as D
from conformer_amd import ops
from conformer_amd.synth import make_state_dict
from oracle import fc_oracle as O

pytestmark = pytest.mark.gpu

TIE = 2e-5       # fp32 logits of magnitude <= 15 summed in a different order differ by about 1e-6


def _model(config, sd):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    m = fc.create_model(config)
    m.load_state_dict(sd)
    return m.cuda().eval()


@pytest.mark.parametrize("name", ["mini_local", "mini_full", "mini_nolen", "small17"])
def test_tokens_of_reference_encoder_output(name):
    """Reference encoder output (fixture) -> device decode == the reference loop's tokens (fixture)."""
    meta, arr = load_golden(name)
    assert meta["frame_stride"] == 1
    m = _model(meta["config"], make_state_dict(meta["config"], meta["weight_seed"]))
    enc = torch.from_numpy(arr["out"]).cuda()
    got = D.greedy_decode(m, enc, max_symbols=10)
    want_counts, want = arr["token_counts"].tolist(), arr["tokens"].tolist()
    same = [len(u) for u in got] == want_counts and [t for u in got for t in u] == want
    assert same or meta["min_logit_gap"] < TIE, (name, meta["min_logit_gap"])


def test_device_form_and_cached_tables():
    meta, arr = load_golden("mini_full")
    m = _model(meta["config"], make_state_dict(meta["config"], meta["weight_seed"]))
    enc = torch.from_numpy(arr["out"]).cuda()
    tokens, counts, last = D.greedy_decode_device(m, enc, max_symbols=10)
    assert last.dtype == torch.int32 and last.shape == (3,)
    assert tokens.dtype == torch.int32 and counts.dtype == torch.int32 and tokens.shape == (3, 26 * 10)
    assert counts.cpu().tolist() == arr["token_counts"].tolist()
    t1 = m._fcenc_joint_tables[1]
    D.greedy_decode_device(m, enc, max_symbols=10)
    assert m._fcenc_joint_tables[1] is t1                         # tables reused
    with torch.no_grad():
        m.joint.output_proj.bias[0] += 100.0                      # blank always wins now
    assert D.greedy_decode(m, enc, max_symbols=10) == [[], [], []]


@pytest.mark.parametrize("vocab,dj,max_symbols", [(29, 320, 3), (29, 512, 3), (29, 640, 2), (32, 1000, 2), (17, 100, 4),
                                                   (128, 256, 1), (1024, 640, 4), (4096, 1024, 2), (2, 1, 5)])
def test_tables_against_oracle(vocab, dj, max_symbols):
    """The kernel alone on random tables: weights in LDS (small V*dj) and streamed from L2 (large)."""
    g = torch.Generator().manual_seed(vocab * 7 + dj)
    B, T = 5, 23
    enc_proj = torch.randn(B, T, dj, generator=g)
    pred_proj = torch.randn(vocab, dj, generator=g)
    out_w = torch.randn(vocab, dj, generator=g) / max(dj, 1) ** 0.5
    out_b = torch.randn(vocab, generator=g) * 0.1
    out_b[0] += 1.0 if vocab > 2 else 0.0                          # blank wins often enough to end frames early
    want, gap = O.greedy_decode_tables(enc_proj, pred_proj, out_w, out_b, 0, max_symbols, 15.0)
    tokens, counts, last = ops.greedy_decode(enc_proj.cuda(), pred_proj.cuda(), out_w.cuda(), out_b.cuda(), dj, 0,
                                             max_symbols, 15.0, T * max_symbols)
    counts = counts.cpu().tolist()
    got = [tokens[i, :n].cpu().tolist() for i, n in enumerate(counts)]
    assert got == want or gap < TIE, gap
    if got == want:
        assert last.cpu().tolist() == [u[-1] if u else 0 for u in want]
    assert any(0 < len(u) < T * max_symbols for u in want)        # both loop exits were taken


def test_ties_take_the_smallest_index_and_clamp():
    """All logits above the clamp collapse to the same value: torch.argmax returns the first of them."""
    dj, vocab = 8, 6
    enc_proj = torch.full((1, 2, dj), 5.0)
    pred_proj = torch.zeros(vocab, dj)
    out_w = torch.zeros(vocab, dj)
    out_w[3] = 10.0
    out_w[4] = 20.0                                               # both far above the clamp of 15
    out_b = torch.zeros(vocab)
    want, _ = O.greedy_decode_tables(enc_proj, pred_proj, out_w, out_b, 0, 2, 15.0)
    assert want == [[3, 3, 3, 3]]
    tokens, counts, _ = ops.greedy_decode(enc_proj.cuda(), pred_proj.cuda(), out_w.cuda(), out_b.cuda(), dj, 0, 2, 15.0, 3)
    assert counts.cpu().tolist() == [4]                            # counted beyond the capacity of 3
    assert tokens.cpu().tolist() == [[3, 3, 3]]
    tokens, counts, _ = ops.greedy_decode(enc_proj.cuda(), pred_proj.cuda(), out_w.cuda(), out_b.cuda(), dj, 0, 2, 0.0, 4)
    assert tokens.cpu().tolist() == [[4, 4, 4, 4]]                 # no clamp: the larger row wins


@pytest.mark.parametrize("name,chunk", [("mini_full", 3), ("small17", 7), ("mini_nolen", 1)])
def test_incremental_decode_equals_one_shot(name, chunk):
    """inference.py:379-406: frames arrive chunk by chunk and the last token is carried over.  The loop is causal in
    the frames, so the chunks' tokens concatenated are the tokens of the whole sequence (fixture)."""
    meta, arr = load_golden(name)
    m = _model(meta["config"], make_state_dict(meta["config"], meta["weight_seed"]))
    enc = torch.from_numpy(arr["out"]).cuda()
    dec = D.IncrementalGreedyDecoder(m, enc.shape[0], max_symbols_per_step=10)
    for t0 in range(0, enc.shape[1], chunk):
        new = dec.step(enc[:, t0:t0 + chunk].contiguous())
        assert len(new) == enc.shape[0]
    same = ([len(u) for u in dec.decoded] == arr["token_counts"].tolist()
            and [t for u in dec.decoded for t in u] == arr["tokens"].tolist())
    assert same or meta["min_logit_gap"] < TIE
    dec.reset()
    assert dec.last is None and dec.decoded == [[] for _ in range(enc.shape[0])]


def test_end_to_end_with_hip_encoder():
    """HIP encoder (fp32 mode) -> device decode == fixture tokens."""
    meta, arr = load_golden("small17")
    from conformer_amd.synth import synthetic_mel
    m = _model(meta["config"], make_state_dict(meta["config"], meta["weight_seed"]))
    m.encoder.set_precision("fp32")
    x, ln = synthetic_mel(meta["B"], meta["T"], meta["lengths"], meta["input_seed"])
    with torch.inference_mode():
        out, _, _ = m.encode(x.cuda(), ln.cuda())
        got = D.greedy_decode(m, out, max_symbols=10)
    assert [len(u) for u in got] == arr["token_counts"].tolist()
    assert [t for u in got for t in u] == arr["tokens"].tolist()


def test_rejects_bad_arguments():
    e = torch.zeros(1, 2, 8, device="cuda")
    w = torch.zeros(4, 8, device="cuda")
    b = torch.zeros(4, device="cuda")
    with pytest.raises(RuntimeError):
        ops.greedy_decode(e, w, w, b, 8, 7, 2, 15.0, 4)            # blank outside the vocabulary
    with pytest.raises(ValueError):
        ops.greedy_decode(e, w[:, :4].contiguous(), w, b, 8, 0, 2, 15.0, 4)
    m = fc.create_model({"model": {"d_encoder": 128, "encoder_layers": 1, "encoder_heads": 2, "subsampling_channels": 128}})
    with pytest.raises(RuntimeError):
        D.greedy_decode(m, torch.zeros(1, 2, 128))                 # CPU tensor: no CPU path
