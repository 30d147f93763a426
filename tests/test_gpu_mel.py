"""Mel front-end kernel (`fcenc_mel_spectrogram` through `conformer_amd.frontend.MelSpectrogram`) against the
CPU oracle (`oracle/mel_oracle.py`: torch.stft + the restated torchaudio filter bank; parity unpinned against
torchaudio itself, which is not installed).

Tolerance: fp32 both sides.  The oracle's FFT and the kernel's matrix DFT each carry about 1e-6 of the frame's
largest spectral power, so |hip - ref| <= 2e-5 * max|ref| over the utterance plus 1e-4 relative."""
import pytest
import torch

from conformer_amd.frontend import MelSpectrogram
from oracle import mel_oracle as MO

pytestmark = pytest.mark.gpu


def _close(got, ref):
    tol = 2e-5 * ref.abs().max() + 1e-4 * ref.abs()
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} beyond tolerance, worst {(got - ref).abs().max():.3e} vs max {ref.abs().max():.3e}"


def _waves(B, S, seed, lengths=None):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(S) / 16000.0
    w = torch.randn(B, S, generator=g) * 0.05
    for b in range(B):                       # a few tones so the spectrum has a large dynamic range
        w[b] += 0.5 * torch.sin(2 * torch.pi * (200.0 + 331.0 * b) * t) + 0.1 * torch.sin(2 * torch.pi * 3100.0 * t)
    if lengths is not None:
        for b, n in enumerate(lengths):
            w[b, n:] = 0.0
    return w


@pytest.mark.parametrize("n_fft,win", [(400, None), (512, 400)])
def test_batch_against_oracle(n_fft, win):
    """Ragged batch: reflect padding at every utterance's own end, zeros past its frames, int64 frame counts."""
    lens = [16000, 15999, 12345, 8000, 201 + 160 * 7, 4321]
    w = _waves(len(lens), 16000, 3, lens)
    m = MelSpectrogram(16000, n_fft=n_fft, win_length=win, hop_length=160, n_mels=80).cuda()
    got, frames = m.batch(w.cuda(), torch.tensor(lens))
    ref, ref_frames = MO.batch_mel(w, lens, sample_rate=16000, n_fft=n_fft, win_length=win, hop_length=160, n_mels=80)
    assert frames.dtype == torch.int64 and torch.equal(frames.cpu(), ref_frames)
    assert got.shape == ref.shape == (len(lens), 101, 80)
    got = got.cpu()
    for b, n in enumerate(ref_frames.tolist()):
        _close(got[b, :n], ref[b, :n])
        assert (got[b, n:] == 0).all()


def test_torchaudio_call_shape_and_single_utterance():
    """transform(waveform (1, samples)) -> (1, n_mels, frames), as train.py:122 / inference.py:134 use it."""
    w = _waves(1, 48000 + 77, 4)
    m = MelSpectrogram(sample_rate=16000, n_mels=80, hop_length=160).cuda()
    out = m(w.cuda())
    assert out.shape == (1, 80, 1 + (48000 + 77) // 160)
    assert out.squeeze(0).transpose(0, 1).is_contiguous()          # the (time, n_mels) the encoder wants
    _close(out.cpu(), MO.mel_spectrogram(w, 16000, 400, None, 160, 80))
    out1 = m(w[0].cuda())
    assert out1.shape == (80, out.shape[2]) and torch.equal(out1, out[0])


def test_streaming_chunk_and_edge_lengths():
    """inference.py:294-300 geometry on short chunks (2720 samples = 17 frames + 1), and the shortest waveform
    reflect padding allows."""
    m = MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80).cuda()
    w = _waves(9, 2720, 5)
    out = m(w.cuda())
    assert out.shape == (9, 80, 18)
    _close(out.cpu(), MO.mel_spectrogram(w, 16000, 512, 400, 160, 80))
    w = _waves(2, 257, 6)
    _close(m(w.cuda()).cpu(), MO.mel_spectrogram(w, 16000, 512, 400, 160, 80))
    with pytest.raises(RuntimeError):
        m(w[:, :256].cuda())


def test_full_size_batch_properties():
    """64 utterances x 20 s (the benchmark batch): too long for a per-utterance oracle loop in seconds as a whole,
    so utterances are spot-checked and the rest is held to properties: a slice of the batch gives the same bits,
    scaling the waveform by 2 scales the mel by exactly 4 (powers of two commute with fp32 rounding)."""
    B, S = 64, 320000
    g = torch.Generator().manual_seed(7)
    lens = torch.randint(S // 2, S + 1, (B,), generator=g)
    lens[0] = S
    w = torch.randn(B, S, generator=g) * 0.1
    m = MelSpectrogram(16000, n_mels=80, hop_length=160).cuda()
    wc = w.cuda()
    full, frames = m.batch(wc, lens)
    assert full.shape == (B, 2001, 80) and torch.isfinite(full).all()
    assert torch.equal(frames.cpu(), 1 + lens // 160)
    part, pf = m.batch(wc[10:13], lens[10:13])
    assert torch.equal(part, full[10:13]) and torch.equal(pf, frames[10:13])
    twice, _ = m.batch(wc * 2.0, lens)
    assert torch.equal(twice, full * 4.0)
    for b in (0, 41):
        n = int(lens[b])
        ref = MO.mel_spectrogram(w[b, :n], 16000, 400, None, 160, 80).transpose(0, 1)
        _close(full[b, :ref.shape[0]].cpu(), ref)
        assert (full[b, ref.shape[0]:] == 0).all()


def test_feeds_the_encoder():
    """waveform -> mel kernel -> HIP encoder, against the oracles chained the same way."""
    import conformer_amd as fc
    from conformer_amd.synth import make_state_dict
    from oracle import fc_oracle as O
    config = {"model": {"d_encoder": 128, "encoder_layers": 2, "encoder_heads": 2, "subsampling_channels": 128}}
    sd = make_state_dict(config, 77)
    model = fc.create_model(config)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    lens = [6400, 5000, 3333]
    w = _waves(3, 6400, 8, lens)
    mel = MelSpectrogram(16000, n_mels=80, hop_length=160).cuda()
    with torch.inference_mode():
        x, frames = mel.batch(w.cuda(), torch.tensor(lens))
        out, out_len, _ = model.encode(x, frames)
    rx, rframes = MO.batch_mel(w, lens, sample_rate=16000, n_fft=400, hop_length=160, n_mels=80)
    ref, ref_len, _ = O.encoder_forward(sd, config, rx, rframes)
    assert torch.equal(out_len.cpu(), ref_len)
    assert (out.cpu() - ref).abs().max() <= 1e-3      # log(mel + 1e-6) amplifies the front-end's 1e-6-level differences


@pytest.mark.parametrize("n_fft,win,S", [(400, None, 2000), (512, 400, 2720), (400, 320, 5100)])
def test_short_ragged_utterances_share_a_tile(n_fft, win, S):
    """At most 32 frames per utterance: several utterances are packed into one 64-frame tile, each reflected at its
    own ends and zero past its own frames; the last tile is partly empty."""
    B = 11
    g = torch.Generator().manual_seed(S)
    lens = torch.randint(n_fft // 2 + 1, S + 1, (B,), generator=g).tolist()
    lens[0], lens[5] = S, n_fft // 2 + 1
    w = _waves(B, S, 9, lens)
    m = MelSpectrogram(16000, n_fft=n_fft, win_length=win, hop_length=160, n_mels=80).cuda()
    got, frames = m.batch(w.cuda(), torch.tensor(lens))
    ref, ref_frames = MO.batch_mel(w, lens, sample_rate=16000, n_fft=n_fft, win_length=win, hop_length=160, n_mels=80)
    assert torch.equal(frames.cpu(), ref_frames) and got.shape == ref.shape
    got = got.cpu()
    for b, n in enumerate(ref_frames.tolist()):
        _close(got[b, :n], ref[b, :n])
        assert (got[b, n:] == 0).all()
