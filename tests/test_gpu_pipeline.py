"""The reference's streaming loop (inference.py:322-376 `process_chunk`: mel of the chunk with n_fft 512 /
win 400, cached encoder step, incremental greedy decode) for several concurrent streams, every stage on the
device: `MelSpectrogram.batch` -> `model.encode(..., cache, streaming=True)` -> `IncrementalGreedyDecoder.step`,
against the CPU oracles chained the same way (mel oracle per stream and chunk, encoder oracle with the
reference's cat / trim caches, greedy loop over the concatenated frames).

fp32 mode.  The mel front-end differs from its oracle at the 1e-6 level and the encoder's log(mel + 1e-6)
amplifies that, so encoder outputs are held to 2e-3 max-abs here (1e-4 is the gate with identical mel input,
tests/test_gpu_streaming.py); tokens must be identical unless a decision was a near-tie."""
import pytest
import torch

import conformer_amd as fc
from conformer_amd import decode as D
from conformer_amd.frontend import MelSpectrogram
from conformer_amd.synth import make_state_dict
from oracle import fc_oracle as O
from oracle import mel_oracle as MO

pytestmark = pytest.mark.gpu


def test_streams_from_waveform_to_tokens():
    config = {"model": {"d_encoder": 128, "encoder_layers": 3, "encoder_heads": 2, "subsampling_channels": 128,
                        "local_context": 16}}
    sd = make_state_dict(config, 91)
    model = fc.create_model(config)
    model.load_state_dict(sd)
    model.encoder.set_precision("fp32")
    model = model.cuda().eval()
    streams, chunk_samples, n_chunks = 5, 2560, 9            # 160 ms chunks -> 17 mel frames -> 3 encoder frames
    g = torch.Generator().manual_seed(92)
    t = torch.arange(chunk_samples * n_chunks) / 16000.0
    audio = torch.randn(streams, chunk_samples * n_chunks, generator=g) * 0.05
    for s in range(streams):
        audio[s] += 0.3 * torch.sin(2 * torch.pi * (300.0 + 170.0 * s) * t)
    mel = MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80).cuda()
    dec = D.IncrementalGreedyDecoder(model, streams, max_symbols_per_step=10)

    cache, ref_cache, outs, ref_outs = None, None, [], []
    with torch.inference_mode():
        for c in range(n_chunks):
            chunk = audio[:, c * chunk_samples:(c + 1) * chunk_samples].contiguous()
            x, frames = mel.batch(chunk.cuda())
            assert x.shape == (streams, 17, 80) and frames.tolist() == [17] * streams
            enc, _, cache = model.encode(x, None, cache=cache, streaming=True)
            dec.step(enc)
            outs.append(enc.cpu())
            rx = MO.mel_spectrogram(chunk, 16000, 512, 400, 160, 80).transpose(1, 2)
            ref, _, ref_cache = O.encoder_forward(sd, config, rx, None, cache=ref_cache, streaming=True)
            ref_outs.append(ref)
    out, ref = torch.cat(outs, 1), torch.cat(ref_outs, 1)
    assert out.shape == ref.shape == (streams, 3 * n_chunks, 128)
    assert (out - ref).abs().max() <= 2e-3
    want, gap = O.greedy_decode(sd, config, ref, max_symbols=10)
    assert dec.decoded == want or gap < 1e-3, gap
