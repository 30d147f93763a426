"""Each HIP kernel, called through the C ABI (conformer_amd.ops -> libfcenc_hip.so), against a plain
fp32 PyTorch CPU computation of the same reference op on identical (seeded) inputs.

Tolerances: fp32 mode 2e-5 relative to the output scale (exact-fp32 MFMA, different summation
order); bf16 mode 2e-2 relative (inputs are rounded to bf16 before BOTH sides compute, so the
difference is accumulation order + one output rounding, and 2e-2 is the budget stated for the
whole encoder)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _tol(dt):
    return 2e-5 if dt == torch.float32 else 2e-2


def _close(got, want, dt, what=""):
    got, want = got.float().cpu(), want.float()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.isfinite(got).all(), what
    err = (got - want).abs().max().item()
    scale = want.abs().max().item() + 1e-6
    assert err <= _tol(dt) * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _r(shape, seed, dt, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("m,n,k", [(300, 256, 256), (128, 128, 64), (1000, 384, 1536), (64, 1024, 256)])
def test_linear_epilogues(dt, m, n, k):
    from conformer_amd import ops, _native as N
    dev = _dev()
    a, w = _r((m, k), 1, dt), _r((n, k), 2, dt, 1 / math.sqrt(k))
    bias = _r((n,), 3, torch.float32)
    ref = a.float() @ w.float().t() + bias
    _close(ops.linear(a.to(dev), w.to(dev), bias.to(dev), N.EPI_STORE), ref, dt, "store")
    _close(ops.linear(a.to(dev), w.to(dev), None, N.EPI_STORE), ref - bias, dt, "store/nobias")
    _close(ops.linear(a.to(dev), w.to(dev), bias.to(dev), N.EPI_STORE_SILU), F.silu(ref), dt, "silu")
    # GLU: the kernel expects rows interleaved in groups of 16 (a | gate)
    half = n // 2
    idx = torch.cat([torch.cat([torch.arange(16 * q, 16 * q + 16), half + torch.arange(16 * q, 16 * q + 16)])
                     for q in range(half // 16)])
    glu_ref = ref[:, :half] * torch.sigmoid(ref[:, half:])
    _close(ops.linear(a.to(dev), w[idx].contiguous().to(dev), bias[idx].contiguous().to(dev), N.EPI_GLU),
           glu_ref, dt, "glu")
    # residual: x = beta x + alpha (a w^T + b), with a compute-dtype copy
    x0 = _r((m, n), 4, torch.float32)
    x = x0.clone().to(dev)
    xc = torch.empty((m, n), dtype=dt, device=dev)
    ops.linear_residual_(x, a.to(dev), w.to(dev), bias.to(dev), 0.5, 1.0, xc)
    _close(x, x0 + 0.5 * ref, dt, "resid")
    _close(xc, x0 + 0.5 * ref, dt, "resid copy")
    x = torch.full((m, n), float("nan"), device=dev)
    ops.linear_residual_(x, a.to(dev), w.to(dev), bias.to(dev), 2.0, 0.0, None)   # beta = 0 must not read x
    _close(x, 2.0 * ref, dt, "resid beta0")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("b,h,w,c,pad", [(2, 21, 10, 128, 0), (1, 8, 40, 256, 0), (3, 13, 5, 128, 64)])
def test_conv3x3s2_bn_silu(dt, b, h, w, c, pad):
    from conformer_amd import ops
    dev = _dev()
    x = _r((b, c, h, w), 5, dt)
    wt = _r((c, c, 3, 3), 6, dt, 1 / math.sqrt(9 * c))
    bias = _r((c,), 7, torch.float32)
    ref = F.silu(F.conv2d(x.float(), wt.float(), bias, stride=2, padding=1)).permute(0, 2, 3, 1)
    got = ops.conv3x3s2_bn_silu(x.permute(0, 2, 3, 1).contiguous().to(dev),
                                wt.permute(0, 2, 3, 1).reshape(c, 9 * c).contiguous().to(dev), bias.to(dev), pad)
    hout = (h + 1) // 2
    assert got.shape[1] == max(pad, hout)
    _close(got[:, :hout], ref, dt, "conv")
    if pad > hout:
        assert float(got[:, hout:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("b,t,c", [(3, 203, 128), (2, 64, 256), (3, 203, 256), (1, 7, 256)])
def test_lognorm_conv1(dt, b, t, c):
    from conformer_amd import ops
    from conformer_amd.synth import synthetic_mel
    from oracle import fc_oracle as O
    dev = _dev()
    lens = [t, max(1, t // 2), max(1, t - 7)][:b]
    mel, _ = synthetic_mel(b, t, lens, 8)
    w = _r((c, 1, 3, 3), 9, torch.float32, 0.3)
    bias = _r((c,), 10, torch.float32, 0.1)
    ref = F.silu(F.conv2d(O.input_norm(mel).unsqueeze(1), w, bias, stride=2, padding=1)).permute(0, 2, 3, 1)
    got = ops.lognorm_conv1(mel.to(dev), w.reshape(c, 9).t().contiguous().to(dev), bias.to(dev),
                            dt == torch.bfloat16)
    # fp32: log/standardise in fp32 on both sides; bf16 only rounds the output
    _close(got, ref, dt if dt == torch.bfloat16 else torch.float32, "conv1")
    if dt == torch.float32:
        assert (got.cpu() - ref).abs().max().item() <= 5e-5


@pytest.mark.parametrize("b,t", [(3, 203), (2, 96), (1, 7), (2, 1000), (1, 49), (5, 48)])
def test_lognorm_conv12_fused(b, t):
    """conv1 produced inside conv2 (one kernel) against (a) the two kernels it replaces, on the same operands, and
    (b) fp32 PyTorch: F.conv2d twice on the standardised log-mel (model.py:856-857, :703-713).  Frame counts cover
    ragged last tiles (output rows not a multiple of 12), a single output row and odd T."""
    from conformer_amd import ops
    from conformer_amd.synth import synthetic_mel
    from oracle import fc_oracle as O
    dev = _dev()
    c = 256
    lens = [t, max(1, t // 2), max(1, t - 7), t, max(1, t // 3)][:b]
    mel, _ = synthetic_mel(b, t, lens, 8)
    w1 = _r((c, 1, 3, 3), 9, torch.float32, 0.3)
    b1 = _r((c,), 10, torch.float32, 0.1)
    w2 = _r((c, c, 3, 3), 11, torch.bfloat16, 1 / math.sqrt(9 * c))
    b2 = _r((c,), 12, torch.float32, 0.1)
    w9 = w1.reshape(c, 9).t().contiguous().to(dev)
    w2k = w2.permute(0, 2, 3, 1).reshape(c, 9 * c).contiguous().to(dev)           # k = (kh*3 + kw)*C + channel
    got = ops.lognorm_conv12(mel.to(dev), w9, b1.to(dev), w2k, b2.to(dev))
    two = ops.conv3x3s2_bn_silu(ops.lognorm_conv1(mel.to(dev), w9, b1.to(dev), True), w2k, b2.to(dev), 0)
    assert got.shape == two.shape and torch.isfinite(got.float()).all()
    # same bf16 conv1 values and weights on both sides; only the order of the fp32 sums differs
    diff = (got.float() - two.float()).abs().max().item()
    assert diff <= 2e-2 * two.float().abs().max().item(), diff
    assert (got != two).float().mean().item() <= 0.05
    y1 = F.silu(F.conv2d(O.input_norm(mel).unsqueeze(1), w1, b1, stride=2, padding=1))
    ref = F.silu(F.conv2d(y1, w2.float(), b2, stride=2, padding=1)).permute(0, 2, 3, 1)
    _close(got, ref, torch.bfloat16, "conv12")


@pytest.mark.parametrize("b,h,pad", [(3, 51, 0), (2, 48, 0), (1, 1, 0), (2, 250, 256), (1, 13, 64), (4, 97, 0)])
def test_conv3_patch(b, h, pad):
    """The third subsampling convolution with its patch staged in LDS against (a) the implicit GEMM on the same
    operands and (b) fp32 PyTorch F.conv2d + SiLU (model.py:715-719).  Row counts cover ragged last tiles (output rows
    not a multiple of 24), a single row, odd and even heights and padded output pitches."""
    from conformer_amd import ops
    dev = _dev()
    c = 256
    x = _r((b, h, 20, c), 21, torch.bfloat16, 1.0)
    w = _r((c, c, 3, 3), 22, torch.bfloat16, 1 / math.sqrt(9 * c))
    bias = _r((c,), 23, torch.float32, 0.1)
    wk = w.permute(0, 2, 3, 1).reshape(c, 9 * c).contiguous().to(dev)
    got = ops.conv3_patch(x.to(dev), wk, bias.to(dev), pad)
    two = ops.conv3x3s2_bn_silu(x.to(dev), wk, bias.to(dev), pad)
    assert got.shape == two.shape and torch.isfinite(got.float()).all()
    diff = (got.float() - two.float()).abs().max().item()
    assert diff <= 2e-2 * two.float().abs().max().item(), diff
    assert (got != two).float().mean().item() <= 0.05
    hout = (h + 1) // 2
    ref = F.silu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, stride=2, padding=1)).permute(0, 2, 3, 1)
    _close(got[:, :hout], ref, torch.bfloat16, "conv3_patch")
    assert not got[:, hout:].any()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("m,d", [(130, 256), (64, 384), (7, 512), (33, 128)])
def test_layernorm_single_and_chained(dt, m, d):
    from conformer_amd import ops
    dev = _dev()
    x = _r((m, d), 11, torch.float32, 3.0) + 0.5
    g1, b1, g2, b2 = (_r((d,), s, torch.float32) for s in (12, 13, 14, 15))
    r1 = F.layer_norm(x, (d,), g1, b1, 1e-5)
    r2 = F.layer_norm(r1, (d,), g2, b2, 1e-5)
    y1, y2 = ops.layernorm(x.to(dev), g1.to(dev), b1.to(dev), None, None, dt == torch.bfloat16)
    _close(y1, r1, torch.float32, "ln fp32 out")
    _close(y2, r1, dt, "ln compute-dtype out")
    y1, y2 = ops.layernorm(x.to(dev), g1.to(dev), b1.to(dev), g2.to(dev), b2.to(dev), dt == torch.bfloat16)
    _close(y1, r1, torch.float32, "chained first")
    _close(y2, r2, dt, "chained second")


@pytest.mark.parametrize("m", [64, 200, 1000])
@pytest.mark.parametrize("post", [False, True])
@pytest.mark.parametrize("d", [256, 384, 512])
def test_ffn_fused(m, post, d):
    """LN -> Linear -> SiLU -> Linear -> scaled residual (-> closing LN) in one kernel vs fp32 PyTorch
    on the same bf16-rounded weights; the three model widths (small, medium, large)."""
    from conformer_amd import ops
    dev = _dev()
    ff = 4 * d
    x0 = _r((m, d), 30, torch.float32, 2.0) + 0.3
    g, b, pg, pb = (_r((d,), s, torch.float32) for s in (31, 32, 33, 34))
    g, pg = g + 1.0, pg + 1.0
    w1, w2 = _r((ff, d), 35, torch.bfloat16, 1 / 16), _r((d, ff), 36, torch.bfloat16, 1 / 32)
    b1, b2 = _r((ff,), 37, torch.float32, 0.2), _r((d,), 38, torch.float32, 0.2)
    h = F.silu(F.layer_norm(x0, (d,), g, b, 1e-5) @ w1.float().t() + b1)
    ref = x0 + 0.5 * (h @ w2.float().t() + b2)
    if post:
        ref = F.layer_norm(ref, (d,), pg, pb, 1e-5)
    x = x0.clone().to(dev)
    xc = torch.empty((m, d), dtype=torch.bfloat16, device=dev)
    w1p, w2p = ops.ffn_pack(w1.to(dev), w2.to(dev))
    ops.ffn_fused_(x, g.to(dev), b.to(dev), w1p, b1.to(dev), w2p, b2.to(dev),
                   pg.to(dev) if post else None, pb.to(dev) if post else None, 0.5, xc)
    _close(x, ref, torch.bfloat16, "ffn fused")
    _close(xc, ref, torch.bfloat16, "ffn fused copy")
    rel = ((x.cpu() - ref).norm() / ref.norm()).item()
    assert rel < 4e-3, rel          # only the LN output and the hidden chunk are rounded to bf16


@pytest.mark.parametrize("m", [64, 200, 1000])
def test_ffn_qkv_fused(m):
    """The feed-forward kernel's optional tail: the q | k | v projection of the updated rows."""
    from conformer_amd import ops
    dev = _dev()
    d, ff, nq = 256, 1024, 768
    x0 = _r((m, d), 50, torch.float32, 2.0) + 0.3
    g, b = _r((d,), 51, torch.float32) + 1.0, _r((d,), 52, torch.float32)
    w1, w2 = _r((ff, d), 53, torch.bfloat16, 1 / 16), _r((d, ff), 54, torch.bfloat16, 1 / 32)
    b1, b2 = _r((ff,), 55, torch.float32, 0.2), _r((d,), 56, torch.float32, 0.2)
    wq, bq = _r((nq, d), 57, torch.bfloat16, 1 / 16), _r((nq,), 58, torch.float32, 0.2)
    h = F.silu(F.layer_norm(x0, (d,), g, b, 1e-5) @ w1.float().t() + b1)
    ref = x0 + 0.5 * (h @ w2.float().t() + b2)
    ref_q = ref @ wq.float().t() + bq
    x = x0.clone().to(dev)
    w1p, w2p = ops.ffn_pack(w1.to(dev), w2.to(dev))
    q = ops.ffn_qkv_fused_(x, g.to(dev), b.to(dev), w1p, b1.to(dev), w2p, b2.to(dev), 0.5,
                           ops.qkv_pack(wq.to(dev)), bq.to(dev))
    _close(x, ref, torch.bfloat16, "ffn fused")
    _close(q, ref_q, torch.bfloat16, "qkv projection")
    rel = ((q.float().cpu() - ref_q).norm() / ref_q.norm()).item()
    assert rel < 8e-3, rel


@pytest.mark.parametrize("d", [384, 512])
@pytest.mark.parametrize("b,tp", [(1, 64), (3, 128), (5, 192)])
def test_convmod_fused_wide(b, tp, d):
    """The conv-module kernel of the medium / large widths (the module alone) against fp32 PyTorch on the same
    bf16-rounded weights; look-back across tile borders and utterance starts as in the d = 256 test below."""
    from conformer_amd import ops
    dev = _dev()
    x0 = _r((b * tp, d), 40, torch.float32, 2.0) + 0.3
    g, bb = _r((d,), 41, torch.float32) + 1.0, _r((d,), 42, torch.float32)
    w1, b1 = _r((2 * d, d), 43, torch.bfloat16, 1 / 16), _r((2 * d,), 44, torch.float32, 0.2)
    dw, dwb = _r((d, 1, 9), 45, torch.float32, 0.3), _r((d,), 46, torch.float32, 0.1)
    w2, b2 = _r((d, d), 47, torch.bfloat16, 1 / 16), _r((d,), 48, torch.float32, 0.2)
    h = F.glu(F.layer_norm(x0, (d,), g, bb, 1e-5) @ w1.float().t() + b1, dim=-1).view(b, tp, d)
    h = F.silu(F.conv1d(F.pad(h.transpose(1, 2), (8, 0)), dw, dwb, groups=d)).transpose(1, 2).reshape(b * tp, d)
    ref = x0 + h @ w2.float().t() + b2
    idx = torch.cat([torch.cat([torch.arange(16 * q, 16 * q + 16), d + torch.arange(16 * q, 16 * q + 16)])
                     for q in range(d // 16)])
    w1p, w2p = ops.convmod_pack(w1[idx].contiguous().to(dev), w2.to(dev))
    got = ops.convmod_fused(x0.to(dev), g.to(dev), bb.to(dev), w1p, b1[idx].contiguous().to(dev),
                            dw.view(d, 9).t().contiguous().to(dev), dwb.to(dev), w2p, b2.to(dev), tp)
    _close(got, ref, torch.bfloat16, "conv module fused (wide)")
    rel = ((got.cpu() - ref).norm() / ref.norm()).item()
    assert rel < 4e-3, rel


@pytest.mark.parametrize("oproj", [False, True])
@pytest.mark.parametrize("b,tp", [(1, 64), (3, 128), (2, 256), (5, 192)])
def test_convmod_fused(b, tp, oproj):
    """LN -> pointwise(GLU) -> causal depthwise 9 (+folded BN) -> SiLU -> pointwise -> residual in one
    kernel vs fp32 PyTorch on the same bf16-rounded weights; several 64-frame tiles per utterance so
    that the 8-frame look-back crosses tile borders and stops at utterance starts.  With oproj the
    kernel first adds the attention out_proj to its input."""
    from conformer_amd import ops
    dev = _dev()
    d = 256
    x_in = _r((b * tp, d), 40, torch.float32, 2.0) + 0.3
    x0 = x_in
    extra = ()
    if oproj:
        attn = _r((b * tp, d), 140, torch.bfloat16)
        wo, bo = _r((d, d), 141, torch.bfloat16, 1 / 16), _r((d,), 142, torch.float32, 0.2)
        x0 = x_in + attn.float() @ wo.float().t() + bo
        extra = (attn.to(dev), ops.rows_pack(wo.to(dev)), bo.to(dev))
    g, bb = _r((d,), 41, torch.float32) + 1.0, _r((d,), 42, torch.float32)
    w1, b1 = _r((2 * d, d), 43, torch.bfloat16, 1 / 16), _r((2 * d,), 44, torch.float32, 0.2)
    dw, dwb = _r((d, 1, 9), 45, torch.float32, 0.3), _r((d,), 46, torch.float32, 0.1)
    w2, b2 = _r((d, d), 47, torch.bfloat16, 1 / 16), _r((d,), 48, torch.float32, 0.2)
    h = F.glu(F.layer_norm(x0, (d,), g, bb, 1e-5) @ w1.float().t() + b1, dim=-1).view(b, tp, d)
    h = F.silu(F.conv1d(F.pad(h.transpose(1, 2), (8, 0)), dw, dwb, groups=d)).transpose(1, 2).reshape(b * tp, d)
    ref = x0 + h @ w2.float().t() + b2
    idx = torch.cat([torch.cat([torch.arange(16 * q, 16 * q + 16), d + torch.arange(16 * q, 16 * q + 16)])
                     for q in range(d // 16)])
    w1p, w2p = ops.convmod_pack(w1[idx].contiguous().to(dev), w2.to(dev))
    got = ops.convmod_fused(x_in.to(dev), g.to(dev), bb.to(dev), w1p, b1[idx].contiguous().to(dev),
                            dw.view(d, 9).t().contiguous().to(dev), dwb.to(dev), w2p, b2.to(dev), tp, *extra)
    _close(got, ref, torch.bfloat16, "conv module fused")
    rel = ((got.cpu() - ref).norm() / ref.norm()).item()
    assert rel < 4e-3, rel          # only the LN output, the GLU output and the depthwise output are rounded to bf16


@pytest.mark.parametrize("b,tp,tv", [(2, 128, 100), (3, 192, 192)])
def test_block_tail_equals_its_two_kernels(b, tp, tv):
    """fcenc_block_tail (out_proj + conv module + second feed-forward + closing LayerNorm in one launch)
    must reproduce, bit for bit, the conv-module kernel followed by the feed-forward kernel: the second
    half only differs in taking its input rows from registers instead of from memory."""
    import ctypes as C
    from conformer_amd import ops, _native as N
    dev = _dev()
    d, ff = 256, 1024
    m = b * tp
    x = (_r((m, d), 60, torch.float32, 2.0) + 0.3).to(dev)
    attn = _r((m, d), 61, torch.bfloat16).to(dev)
    idx = torch.cat([torch.cat([torch.arange(16 * q, 16 * q + 16), d + torch.arange(16 * q, 16 * q + 16)])
                     for q in range(d // 16)])
    t = {}
    t["conv_ln_g"], t["conv_ln_b"] = (_r((d,), 62, torch.float32) + 1.0).to(dev), _r((d,), 63, torch.float32).to(dev)
    t["pw1_w"], t["pw2_w"] = ops.convmod_pack(_r((2 * d, d), 64, torch.bfloat16, 1 / 16)[idx].contiguous().to(dev),
                                              _r((d, d), 65, torch.bfloat16, 1 / 16).to(dev))
    t["pw1_b"], t["pw2_b"] = _r((2 * d,), 66, torch.float32, 0.2).to(dev), _r((d,), 67, torch.float32, 0.2).to(dev)
    t["dw_w9"], t["dw_b"] = _r((9, d), 68, torch.float32, 0.3).to(dev), _r((d,), 69, torch.float32, 0.1).to(dev)
    t["out_w"], t["out_b"] = ops.rows_pack(_r((d, d), 70, torch.bfloat16, 1 / 16).to(dev)), _r((d,), 71, torch.float32, 0.2).to(dev)
    t["ff2_ln_g"], t["ff2_ln_b"] = (_r((d,), 72, torch.float32) + 1.0).to(dev), _r((d,), 73, torch.float32).to(dev)
    t["ff2_w1"], t["ff2_w2"] = ops.ffn_pack(_r((ff, d), 74, torch.bfloat16, 1 / 16).to(dev),
                                            _r((d, ff), 75, torch.bfloat16, 1 / 32).to(dev))
    t["ff2_b1"], t["ff2_b2"] = _r((ff,), 76, torch.float32, 0.2).to(dev), _r((d,), 77, torch.float32, 0.2).to(dev)
    t["ln_g"], t["ln_b"] = (_r((d,), 78, torch.float32) + 1.0).to(dev), _r((d,), 79, torch.float32).to(dev)

    want = ops.convmod_fused(x, t["conv_ln_g"], t["conv_ln_b"], t["pw1_w"], t["pw1_b"], t["dw_w9"], t["dw_b"],
                             t["pw2_w"], t["pw2_b"], tp, attn, t["out_w"], t["out_b"])
    ops.ffn_fused_(want, t["ff2_ln_g"], t["ff2_ln_b"], t["ff2_w1"], t["ff2_b1"], t["ff2_w2"], t["ff2_b2"],
                   t["ln_g"], t["ln_b"], 0.5, None)

    layer = N.Layer()
    for k, v in t.items():
        setattr(layer, k, v.data_ptr())
    got = torch.full((m, d), float("nan"), device=dev)
    compact = torch.full((b, tv, d), float("nan"), device=dev)
    N.check(N.lib().fcenc_block_tail(C.byref(layer), N.ptr(x), N.ptr(got), N.ptr(attn), N.ptr(compact), m, d, ff, 4,
                                     tp, tv, N.stream_of(x)))
    assert torch.equal(got.cpu(), want.cpu())
    assert torch.equal(compact.cpu(), want.view(b, tp, d)[:, :tv].cpu())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("b,t,d", [(2, 64, 256), (3, 128, 384), (1, 192, 512), (2, 64, 128)])
def test_dwconv9_bn_silu(dt, b, t, d):
    from conformer_amd import ops
    dev = _dev()
    x = _r((b, t, d), 16, dt)
    w = _r((d, 1, 9), 17, torch.float32, 0.3)
    bias = _r((d,), 18, torch.float32, 0.1)
    ref = F.silu(F.conv1d(F.pad(x.float().transpose(1, 2), (8, 0)), w, bias, groups=d)).transpose(1, 2)
    got = ops.dwconv9_bn_silu(x.to(dev), w.view(d, 9).t().contiguous().to(dev), bias.to(dev))
    _close(got, ref, dt, "dwconv")


def _attention_reference(qkv, pos, bu, bv, klen, B, H, Tp, Tv, window, scale):
    """Plain fp32 evaluation of model.py:379-428 on the padded-head layout."""
    hd = H * 64
    q, k, v = (qkv.float().view(B, Tp, 3, H, 64)[:, :, i].permute(0, 2, 1, 3) for i in range(3))   # (B,H,Tp,64)
    p = pos.float().view(-1, H, 64).permute(1, 0, 2)                                                # (H,R,64)
    i = torch.arange(Tp).view(-1, 1)
    j = torch.arange(Tp).view(1, -1)
    ac = torch.matmul(q + bu.view(1, H, 1, 64), k.transpose(-1, -2))
    bd_full = torch.matmul(q + bv.view(1, H, 1, 64), p.transpose(-1, -2).unsqueeze(0))              # (B,H,Tp,R)
    rel = ((Tv - 1) + (j - i) + 64).clamp(0, pos.shape[0] - 1).expand(B, H, Tp, Tp)
    bd = torch.gather(bd_full, 3, rel)
    s = (ac + bd) * scale
    dead = j >= klen.view(B, 1, 1, 1)
    if window > 0:
        dead = dead | (j > i) | (j < i - window + 1)
    s = s.masked_fill(dead, float("-inf"))
    return torch.matmul(torch.softmax(s, -1), v).permute(0, 2, 1, 3).reshape(B * Tp, hd)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,H,Tp,Tv,klens,window", [
    (2, 2, 128, 100, [100, 70], 0),
    (1, 4, 64, 64, [64], 0),
    (2, 2, 192, 170, [170, 129], 48),
    (1, 2, 256, 250, [250], 128),
])
def test_relpos_attention(dt, B, H, Tp, Tv, klens, window):
    from conformer_amd import ops
    dev = _dev()
    hd = H * 64
    qkv = _r((B * Tp, 3 * hd), 19, dt)
    pos = torch.zeros((2 * Tp + 64, hd))
    pos[64:64 + 2 * Tv - 1] = _r((2 * Tv - 1, hd), 20, torch.float32)
    pos = pos.to(dt)
    bu, bv = _r((hd,), 21, torch.float32, 0.3), _r((hd,), 22, torch.float32, 0.3)
    klen = torch.tensor(klens, dtype=torch.int32)
    scale = 0.125
    ref = _attention_reference(qkv, pos, bu, bv, klen, B, H, Tp, Tv, window, scale)
    got = ops.relpos_attention(qkv.to(dev), pos.to(dev), bu.to(dev), bv.to(dev), klen.to(dev), B, Tv, window, scale)
    got = got.float().cpu().view(B, Tp, hd)
    ref = ref.view(B, Tp, hd)
    # rows the reference defines: every frame < Tv whose window holds at least one visible key
    for b in range(B):
        rows = torch.arange(Tv)
        if window > 0:
            rows = rows[rows - window + 1 < klens[b]]
        _close(got[b, rows], ref[b, rows], dt, f"attention b={b}")


def test_relpos_attention_rows_without_keys_are_nan():
    """A query whose whole window is masked is NaN in the reference (softmax over -inf); keep that."""
    from conformer_amd import ops
    dev = _dev()
    B, H, Tp, Tv, window = 1, 2, 128, 128, 16
    qkv = _r((B * Tp, 3 * H * 64), 23, torch.bfloat16)
    pos = _r((2 * Tp + 64, H * 64), 24, torch.bfloat16)
    z = torch.zeros(H * 64)
    klen = torch.tensor([40], dtype=torch.int32)
    got = ops.relpos_attention(qkv.to(dev), pos.to(dev), z.to(dev), z.to(dev), klen.to(dev), B, Tv, window, 0.125)
    got = got.float().cpu().view(Tp, -1)
    assert torch.isfinite(got[:55]).all()            # frames 0..54 see key >= i-15 < 40
    assert torch.isnan(got[55:]).all()


@pytest.mark.parametrize("lengths", [None, [1000, 873, 640, 411], [1, 2, 8, 9, 0], [2001, 17]])
def test_output_lengths_and_mask_bit_exact(lengths):
    from conformer_amd import ops
    from oracle import fc_oracle as O
    dev = _dev()
    t_valid = 251
    b = 4 if lengths is None else len(lengths)
    ln = None if lengths is None else torch.tensor(lengths, dtype=torch.int64)
    out_len, mask, klen = ops.output_lengths(None if ln is None else ln.to(dev), b, t_valid, dev)
    ref_len, ref_mask = O.output_lengths_and_mask(ln, b, t_valid)
    assert out_len.dtype == torch.int64 and torch.equal(out_len.cpu(), ref_len)
    if ref_mask is None:
        assert not mask.any()
        assert klen.cpu().tolist() == [t_valid] * b
    else:
        assert mask.dtype == torch.bool and torch.equal(mask.cpu(), ref_mask)
        assert klen.cpu().tolist() == [min(int(v), t_valid) for v in ref_len]


def test_shape_errors_are_reported_not_launched():
    from conformer_amd import ops, _native as N
    dev = _dev()
    a = torch.zeros((64, 100), dtype=torch.bfloat16, device=dev)       # K not a multiple of 64
    w = torch.zeros((128, 100), dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="multiple"):
        ops.linear(a, w, None, N.EPI_STORE)
    with pytest.raises(TypeError):
        ops.linear(a.float(), w, None, N.EPI_STORE)


def test_front_end_kernels_reject_shapes_they_are_not_built_for():
    """The two patch-in-LDS kernels exist for 80 mel bins / 20 conv3 input columns and 256 channels in bf16; anything
    else must come back as an error from the C entry point (nothing launched), and the `*_supported` queries must say
    so beforehand, which is how the encoder pass chooses the general kernels instead."""
    from conformer_amd import ops, _native as N
    dev = _dev()
    lib = N.lib()
    assert lib.fcenc_conv12_supported(N.BF16, 80, 256) == 1
    assert lib.fcenc_conv12_supported(N.F32, 80, 256) == 0
    assert lib.fcenc_conv12_supported(N.BF16, 64, 256) == 0
    assert lib.fcenc_conv12_supported(N.BF16, 80, 128) == 0
    assert lib.fcenc_conv3_patch_supported(N.BF16, 20, 256, 256) == 1
    assert lib.fcenc_conv3_patch_supported(N.BF16, 40, 256, 256) == 0
    assert lib.fcenc_conv3_patch_supported(N.BF16, 20, 128, 128) == 0
    assert lib.fcenc_conv3_patch_supported(N.F32, 20, 256, 256) == 0
    x = torch.zeros((1, 8, 40, 256), dtype=torch.bfloat16, device=dev)          # 40 columns: conv2's shape, not conv3's
    w = torch.zeros((256, 9 * 256), dtype=torch.bfloat16, device=dev)
    bias = torch.zeros((256,), dtype=torch.float32, device=dev)
    with pytest.raises(ValueError, match="20 input columns"):
        ops.conv3_patch(x, w, bias, 0)
    out = torch.zeros((1, 4, 20, 256), dtype=torch.bfloat16, device=dev)
    wp = torch.zeros((lib.fcenc_conv3_patch_packed_bytes(),), dtype=torch.uint8, device=dev)
    rc = lib.fcenc_conv3_patch(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), out.data_ptr(), 1, 8, 40, 256, 4, N.stream_of(x))
    assert rc != 0 and b"conv3p" in lib.fcenc_last_error()
    mel = torch.ones((1, 16, 64), dtype=torch.float32, device=dev)              # 64 mel bins
    with pytest.raises(ValueError, match="80 mel bins"):
        ops.lognorm_conv12(mel, torch.zeros((9, 256), device=dev), bias, w, bias)
    torch.cuda.synchronize()
    assert not out.any()
