"""Drop-in boundary on CPU: state_dict contract, config handling, seeded init identical to the
reference, composite (training/streaming) path equal to the oracle, loud failure of the eval
path without a GPU, and the C ABI exporting every declared symbol."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import REFERENCE, ROOT, load_golden
import conformer_amd as fc
from conformer_amd.synth import make_state_dict, state_dict_shapes, synthetic_mel
from oracle import fc_oracle as O

SMALL = {"model": {"d_encoder": 256, "encoder_layers": 17, "encoder_heads": 4}}


def test_state_dict_contract_small():
    m = fc.create_model(SMALL)
    sd = m.state_dict()
    want = state_dict_shapes(SMALL)
    assert len(sd) == 683                                   # reference count at 17 layers
    assert list(sd.keys()) == list(want.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(want[k]), k
    assert sd["encoder.pos_encoding.pe"].shape == (9999, 256)


def test_checkpoint_roundtrip_like_inference_loader(tmp_path):
    """train.py:403-413 writes {'model_state_dict': ...}; inference.py:553-554 loads it."""
    meta, _ = load_golden("mini_full")
    sd = make_state_dict(meta["config"], 3)
    path = tmp_path / "best_model.pt"
    torch.save({"epoch": 1, "wer": 1.0, "model_state_dict": sd, "config": meta["config"]}, path)
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    m = fc.create_model(ckpt["config"])
    m.load_state_dict(ckpt["model_state_dict"])
    m.eval()
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k


@pytest.mark.parametrize("name", ["mini_local", "mini_nolen"])
def test_composite_training_path_matches_fixture(name):
    """The autograd/streaming composite (not the shipped eval path) agrees with the reference's numbers."""
    meta, arr = load_golden(name)
    m = fc.create_model(meta["config"])
    m.load_state_dict(make_state_dict(meta["config"], meta["weight_seed"]))
    m.eval()
    x, ln = synthetic_mel(meta["B"], meta["T"], meta["lengths"], meta["input_seed"])
    with torch.no_grad():
        out, out_len, _ = m.encoder.forward_composite(x, ln)
    assert np.array_equal(out_len.numpy(), arr["out_len"])
    assert np.abs(out.numpy() - arr["out"]).max() <= 1e-5


def test_composite_streaming_matches_fixture():
    meta, arr = load_golden("mini_streaming")
    m = fc.create_model(meta["config"])
    m.load_state_dict(make_state_dict(meta["config"], meta["weight_seed"]))
    m.eval()
    x, _ = synthetic_mel(meta["B"], sum(meta["chunk_frames"]), None, meta["input_seed"])
    cache, pos = None, 0
    with torch.no_grad():
        for i, cf in enumerate(meta["chunk_frames"]):
            out, _, cache = m.encode(x[:, pos:pos + cf], None, cache=cache, streaming=True)
            pos += cf
            assert np.abs(out.numpy() - arr[f"out_{i}"]).max() <= 1e-5, i
    init = m.get_initial_cache(2, torch.device("cpu"))
    assert len(init) == 2 and init[0][0] is None and tuple(init[0][1].shape) == (2, 128, 8)


def test_training_mode_has_gradients():
    meta, _ = load_golden("mini_full")
    torch.manual_seed(0)
    m = fc.create_model(meta["config"]).train()
    for blk in m.encoder.layers:
        blk.drop_prob = 0.0          # stochastic depth (model.py:659) would skip a block now and then: not under test here
    x, ln = synthetic_mel(2, 96, [96, 64], 5)
    tgt = torch.randint(1, 29, (2, 5))
    logits, enc_len = m(x, ln, tgt, torch.tensor([5, 4]))
    assert logits.shape[:3] == (2, 12, 6) and logits.shape[3] == 29
    logits.float().pow(2).mean().backward()
    assert m.encoder.layers[0].ff1.linear1.weight.grad is not None


def test_eval_path_refuses_cpu():
    meta, _ = load_golden("mini_full")
    m = fc.create_model(meta["config"]).eval()
    x, ln = synthetic_mel(1, 64, None, 1)
    with pytest.raises(RuntimeError, match="MI355X"):
        m.encode(x, ln)


def test_stochastic_depth_schedule():
    assert fc.stochastic_depth_schedule(4, 0.2, "linear", 1) == pytest.approx([0.05, 0.1, 0.15, 0.2])
    assert fc.stochastic_depth_schedule(4, 0.2, "uniform", 3) == [0.0, 0.0, 0.2, 0.2]
    with pytest.raises(ValueError):
        fc.stochastic_depth_schedule(4, 1.0)
    with pytest.raises(ValueError):
        fc.stochastic_depth_schedule(4, 0.1, "cosine")


def test_c_abi_exports_every_declared_symbol():
    """Loads the library (no GPU work) and checks every function of include/fcenc.h is exported."""
    from conformer_amd import _native
    header = open(os.path.join(ROOT, "include", "fcenc.h")).read()
    declared = set(re.findall(r"\b(fcenc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_native.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _native.lib().fcenc_version() >= 100
    assert _native.lib().fcenc_out_frames(2000) == 250 and _native.lib().fcenc_padded_frames(2000) == 256


@pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "model.py")), reason="reference not mounted")
def test_seeded_init_identical_to_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REFERENCE)
    import model as ref_model
    cfg = json.load(open(os.path.join(REFERENCE, "config", "fastconformer_medium.json")))
    cfg["model"]["encoder_layers"] = 2
    torch.manual_seed(123)
    ref = ref_model.create_model(cfg).state_dict()
    torch.manual_seed(123)
    mine = fc.create_model(cfg).state_dict()
    assert list(ref.keys()) == list(mine.keys())
    for k in ref:
        assert torch.equal(ref[k], mine[k]), k


def test_whole_pass_operators_trace_under_fake_tensors():
    """`model.encode` is one torch.library operator (fcenc::encoder_forward): under FakeTensorMode -- what torch.compile
    and export use to propagate shapes, and what runs without a GPU -- it yields the reference's shapes, dtypes and
    device (model.py:835-894: (B, T', d) float32, (B,) int64, no cache), with and without lengths, under
    inference_mode (inference.py:111) and autocast (train.py:301)."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from conformer_amd import ops
    cfg = fc.load_config(os.path.join(ROOT, "configs", "fastconformer_medium.json"))
    m = fc.create_model(cfg).eval()
    mel = fc.frontend.MelSpectrogram(sample_rate=16000, n_mels=80, hop_length=160)
    with FakeTensorMode(allow_non_fake_inputs=True):
        x = torch.empty((4, 1003, 80), device="cuda")
        for ln in (torch.tensor([1003, 873, 640, 411]), None):
            with torch.inference_mode():
                out, out_len, cache = m.encode(x, ln)
            assert tuple(out.shape) == (4, 126, 384) and out.dtype == torch.float32 and out.device.type == "cuda"
            assert tuple(out_len.shape) == (4,) and out_len.dtype == torch.int64 and cache is None
        # the cached step and the mel front-end as operators
        k = [torch.empty((4, 8, 256, 64), device="cuda", dtype=torch.bfloat16)]
        v = [torch.empty((4, 8, 32, 64, 8), device="cuda", dtype=torch.bfloat16)]
        c = [torch.empty((4, 8, 384), device="cuda", dtype=torch.bfloat16)]
        y = torch.ops.fcenc.encoder_stream_step(torch.empty((4, 17, 80), device="cuda"), k, v, c, 0, 0, 384, 1, False)
        assert tuple(y.shape) == (4, 3, 384) and y.dtype == torch.float32
        mel_x, frames = torch.ops.fcenc.mel_spectrogram(
            torch.empty((2, 16000), device="cuda"), None, mel._dft, mel._win, mel._fb_start, mel._fb_count, mel._fb_w,
            400, 160, mel._scalars["kp"], mel._scalars["wpitch"], 80, mel._scalars["fb_maxw"])
        assert tuple(mel_x.shape) == (2, 101, 80) and tuple(frames.shape) == (2,) and frames.dtype == torch.int64
    # outside fake mode a dead handle is an error, never a silent no-op
    with pytest.raises(RuntimeError, match="handle"):
        ops.from_handle(10 ** 9)
    schema = str(torch.ops.fcenc.encoder_stream_step.default._schema)
    assert "Tensor(a2!)[] kcache" in schema or "(a" in schema        # the rings are declared as mutated


def test_packed_weights_are_dropped_when_weights_change():
    """What decides a re-pack, without a GPU: the version tuple moves with any in-place write to a parameter or a
    buffer, train() drops what was packed, and packed() compares versions before it reuses anything."""
    cfg = {"model": {"d_encoder": 128, "encoder_layers": 2, "encoder_heads": 2, "subsampling_channels": 128}}
    enc = fc.create_model(cfg).encoder.eval()
    v0 = enc._weights_version()
    assert enc._weights_version() == v0
    with torch.no_grad():
        enc.layers[1].ff1.linear1.weight.mul_(1.01)
    v1 = enc._weights_version()
    assert v1 != v0
    enc.layers[0].conv.norm.running_mean.add_(0.5)          # a buffer, as a BatchNorm forward updates it
    assert enc._weights_version() != v1
    sentinel = object()
    enc._packed[("cuda:0", torch.float32)] = sentinel
    enc._packed_version = enc._weights_version()
    assert enc.packed(torch.device("cuda:0"), torch.float32) is sentinel     # nothing changed: reused
    enc.train()
    assert enc._packed == {}
    enc.eval()
    enc._packed[("cuda:0", torch.float32)] = sentinel
    enc._packed_version = enc._weights_version()
    with torch.no_grad():
        enc.subsampling.linear.bias.add_(1.0)
    with pytest.raises(Exception):            # stale entry dropped; re-packing needs the GPU, which is absent here
        enc.packed(torch.device("cuda:0"), torch.float32)
    assert enc._packed == {}


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No silent fallback: without the built library every product entry raises and says how to build it."""
    from conformer_amd import _native
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "libfcenc_hip.so"))
    monkeypatch.setattr(_native, "_lib", None)
    with pytest.raises(Exception, match="conformer_amd.build"):
        _native.lib()
