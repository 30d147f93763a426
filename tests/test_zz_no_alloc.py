"""The one entry point of the library that allocates nothing (`fcenc_lognorm_conv1_static`: input statistics +
conv1 with the partial sums in a fixed device array of the library) against the two-call form it mirrors
(`fcenc_input_stats` into a caller buffer, then `fcenc_conv1`), through the C ABI, bit for bit; two calls in a
row, the second with a smaller input, so that whatever the first call left in the array is shown not to matter."""
import pytest
import torch

from conformer_amd import _native as N
from conformer_amd.synth import synthetic_mel

pytestmark = pytest.mark.gpu


def _two_call_form(dt, mel, w9, bias, c):
    b, t, f = mel.shape
    lib, s = N.lib(), N.stream_of(mel)
    partial = torch.empty((b, 16, 2), dtype=torch.float64, device=mel.device)
    out = torch.empty((b, (t + 1) // 2, (f + 1) // 2, c), dtype=dt, device=mel.device)
    N.check(lib.fcenc_input_stats(N.ptr(mel), N.ptr(partial), b, t, f, 16, s))
    N.check(lib.fcenc_conv1(N.dtype_code(dt), N.ptr(mel), N.ptr(partial), 16, N.ptr(w9), N.ptr(bias), N.ptr(out), b, t, f, c, s))
    return out


def _static_form(dt, mel, w9, bias, c):
    b, t, f = mel.shape
    out = torch.empty((b, (t + 1) // 2, (f + 1) // 2, c), dtype=dt, device=mel.device)
    N.check(N.lib().fcenc_lognorm_conv1_static(N.dtype_code(dt), N.ptr(mel), N.ptr(w9), N.ptr(bias), N.ptr(out), b, t, f, c,
                                               N.stream_of(mel)))
    return out


@pytest.mark.parametrize("dt,c", [(torch.bfloat16, 256), (torch.float32, 256), (torch.float32, 128)])
def test_static_scratch_entry_point_matches_the_two_call_form(dt, c):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    g = torch.Generator().manual_seed(5)
    w9 = (torch.randn((9, c), generator=g) * 0.3).cuda()
    bias = (torch.randn((c,), generator=g) * 0.1).cuda()
    # a larger batch first, then a smaller one with other data: the array keeps the first call's sums beyond row 3
    for batch, frames, lens, seed in ((7, 403, [403, 300, 77, 403, 11, 250, 399], 1), (3, 120, [120, 64, 9], 2)):
        mel, _ = synthetic_mel(batch, frames, lens, seed)
        mel = mel.cuda()
        want = _two_call_form(dt, mel, w9, bias, c)
        got = _static_form(dt, mel, w9, bias, c)
        torch.cuda.synchronize()
        assert torch.isfinite(got.float()).all()
        assert torch.equal(got, want), (batch, frames)


def test_static_scratch_entry_point_rejects_what_does_not_fit():
    mel = torch.ones((1, 8, 80), device="cuda")
    w9, bias, out = torch.zeros((9, 256), device="cuda"), torch.zeros(256, device="cuda"), torch.empty(4, device="cuda")
    rc = N.lib().fcenc_lognorm_conv1_static(N.BF16, N.ptr(mel), N.ptr(w9), N.ptr(bias), N.ptr(out), 4097, 8, 80, 256,
                                            N.stream_of(mel))
    assert rc != 0 and b"4096" in N.lib().fcenc_last_error()
