"""Training path (train.py:302-336: model.forward + loss.backward) on the HIP kernels of conformer_amd.autograd,
against ATen: each autograd function alone (outputs and every gradient), then the whole encoder and the whole
transducer forward in training mode with `native_training` on and off from the same weights -- outputs, gradients
of every parameter and of the input, and the BatchNorm running statistics left behind.  fp32 throughout (the
training default; the reference's fp32 gate of 1e-4 is applied relative to each tensor's largest element),
then AMP: bf16 operands with fp32 accumulation, pinned against an emulation of exactly that and compared with ATen's autocast."""
import pytest
import torch
import torch.nn.functional as F

import conformer_amd as fc
from conformer_amd import autograd as A
from conformer_amd.synth import make_state_dict, synthetic_mel

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _close(a, b, what, tol=TOL, floor=1e-6):
    """|a - b| <= tol x the reference's largest element (or `floor`: a convolution bias in front of a training-mode
    BatchNorm has an exactly zero gradient, of which both paths only hold rounding noise)."""
    assert a.shape == b.shape, what
    assert torch.isfinite(a).all(), what
    scale = max(b.abs().max().item(), floor)
    err = (a - b).abs().max().item() / scale
    assert err <= tol, f"{what}: {err:.3e} of the largest element"


@pytest.mark.parametrize("m,k,n", [(200, 256, 1024), (77, 1024, 256), (640, 128, 384), (64, 2560, 256)])
def test_linear_forward_and_both_gradients(m, k, n):
    g = torch.Generator().manual_seed(m + n)
    shape = (4, m // 4, k) if m % 4 == 0 else (m, k)          # batched and flat inputs; token counts off the 64 grid
    x = torch.randn(shape, generator=g).cuda().requires_grad_()
    w = (torch.randn((n, k), generator=g) * 0.05).cuda().requires_grad_()
    b = torch.randn(n, generator=g).cuda().requires_grad_()
    dy = torch.randn(x.shape[:-1] + (n,), generator=g).cuda()
    y = A.linear(x, w, b)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), dy)
    xr, wr, br = (t.detach().double().requires_grad_() for t in (x, w, b))
    yr = F.linear(xr, wr, br)
    rx, rw, rb = torch.autograd.grad(yr, (xr, wr, br), dy.double())
    for got, want, nm in ((y, yr, "y"), (gx, rx, "dx"), (gw, rw, "dW"), (gb, rb, "db")):
        _close(got.detach().double(), want.detach(), f"linear {nm} {m}x{k}x{n}", 2e-5)
    # without bias, input not requiring a gradient
    y2 = A.linear(x.detach(), w, None)
    (gw2,) = torch.autograd.grad(y2, (w,), dy)
    _close(gw2.double(), rw, "dW without bias", 2e-5)


@pytest.mark.parametrize("m,k,n", [(256, 128, 512), (40, 1280, 128), (39, 128, 128)])
def test_linear_bf16_is_the_exact_product_of_rounded_operands(m, k, n):
    """AMP mode: operands rounded to bf16, products and sums in fp32, fp32 results -- forward, dx and dW."""
    g = torch.Generator().manual_seed(m + k)
    x = torch.randn((m, k), generator=g).cuda().requires_grad_()
    w = (torch.randn((n, k), generator=g) * 0.05).cuda().requires_grad_()
    dy = torch.randn((m, n), generator=g).cuda()
    y = A.linear(x, w, None, True)
    assert y.dtype == torch.float32
    gx, gw = torch.autograd.grad(y, (x, w), dy)
    xb, wb, dyb = (t.detach().bfloat16().double() for t in (x, w, dy))
    for got, want, nm in ((y, xb @ wb.t(), "y"), (gx, dyb @ wb, "dx"), (gw, dyb.t() @ xb, "dW")):
        _close(got.detach().double(), want, f"bf16 linear {nm} {m}x{k}x{n}", 2e-6)


@pytest.mark.parametrize("d", [128, 256, 384, 512])
def test_layer_norm_forward_and_gradients(d):
    g = torch.Generator().manual_seed(d)
    x = (torch.randn((3, 37, d), generator=g) * 2 + 0.5).cuda().requires_grad_()
    w = (1 + 0.3 * torch.randn(d, generator=g)).cuda().requires_grad_()
    b = (0.2 * torch.randn(d, generator=g)).cuda().requires_grad_()
    dy = torch.randn((3, 37, d), generator=g).cuda()
    y = A.layer_norm(x, w, b)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), dy)
    xr, wr, br = (t.detach().double().requires_grad_() for t in (x, w, b))
    yr = F.layer_norm(xr, (d,), wr, br, 1e-5)
    rx, rw, rb = torch.autograd.grad(yr, (xr, wr, br), dy.double())
    for got, want, nm in ((y, yr, "y"), (gx, rx, "dx"), (gw, rw, "dweight"), (gb, rb, "dbias")):
        _close(got.detach().double(), want.detach(), f"layer_norm {nm} d={d}", 2e-5)


@pytest.mark.parametrize("b,t,d", [(2, 50, 128), (3, 16, 256), (1, 7, 384), (2, 129, 512)])
def test_glu_dwconv_forward_and_gradients(b, t, d):
    g = torch.Generator().manual_seed(t + d)
    y = torch.randn((b, t, 2 * d), generator=g).cuda().requires_grad_()
    w = (torch.randn((d, 1, 9), generator=g) * 0.3).cuda().requires_grad_()
    bias = (torch.randn(d, generator=g) * 0.1).cuda().requires_grad_()
    dz = torch.randn((b, t, d), generator=g).cuda()
    z = A.glu_dwconv(y, w, bias)
    gy, gw, gb = torch.autograd.grad(z, (y, w, bias), dz)
    yr, wr, br = (v.detach().double().requires_grad_() for v in (y, w, bias))
    glu = F.glu(yr.transpose(1, 2), dim=1)                                     # model.py:521
    zr = F.conv1d(F.pad(glu, (8, 0)), wr, br, groups=d).transpose(1, 2)        # model.py:236-249
    ry, rw, rb = torch.autograd.grad(zr, (yr, wr, br), dz.double())
    for got, want, nm in ((z, zr, "z"), (gy, ry, "dy"), (gw, rw, "dweight"), (gb, rb, "dbias")):
        _close(got.detach().double(), want.detach(), f"glu_dwconv {nm} {b}x{t}x{d}", 2e-5)


def _train_model(config, seed, native):
    m = fc.create_model(config)
    m.load_state_dict(make_state_dict(config, seed))
    m = m.cuda().train()
    m.encoder.native_training = native
    return m


CFG = {"model": {"d_encoder": 128, "encoder_layers": 3, "encoder_heads": 2, "subsampling_channels": 128,
                 "local_context": 16, "dropout": 0.0, "stochastic_depth_prob": 0.0}}


@pytest.mark.parametrize("shape", [(2, 50, 128), (3, 41, 256), (5, 13, 384), (1, 200, 512)])
def test_batchnorm_train_silu_forward_gradients_and_running_statistics(shape):
    """nn.BatchNorm1d with batch statistics + SiLU (model.py:528-537 in training mode), token-major: output, the
    three gradients, running_mean / running_var / num_batches_tracked left behind, against fp64 ATen; the large
    offset of the input makes a one-pass variance (sum of squares minus squared mean) fail the tolerance."""
    d = shape[-1]
    g = torch.Generator().manual_seed(d)
    z = (torch.randn(shape, generator=g) * 1.7 + 30.0).cuda().requires_grad_()
    bn = torch.nn.BatchNorm1d(d).cuda().train()
    ref = torch.nn.BatchNorm1d(d).cuda().double().train()
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * torch.randn(d, generator=g))
        bn.bias.copy_(0.2 * torch.randn(d, generator=g))
        bn.running_mean.copy_(torch.randn(d, generator=g))
        bn.running_var.copy_(torch.rand(d, generator=g) + 0.5)
        ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    dy = torch.randn(shape, generator=g).cuda()
    assert A.bn_silu_ok(bn, d)
    versions = (bn.running_mean._version, bn.running_var._version)
    y = A.bn_silu(z, bn)
    gz, gw, gb = torch.autograd.grad(y, (z, bn.weight, bn.bias), dy)
    zr = z.detach().double().requires_grad_()
    yr = F.silu(ref(zr.reshape(-1, d))).view(shape)
    rz, rw, rb = torch.autograd.grad(yr, (zr, ref.weight, ref.bias), dy.double())
    for got, want, nm in ((y, yr, "y"), (gz, rz, "dz"), (gw, rw, "dweight"), (gb, rb, "dbias"),
                          (bn.running_mean, ref.running_mean, "running_mean"), (bn.running_var, ref.running_var, "running_var")):
        _close(got.detach().double(), want.detach(), f"bn_silu {nm} {shape}", 2e-5)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    assert bn.running_mean._version > versions[0] and bn.running_var._version > versions[1]
    bn.eval()
    assert not A.bn_silu_ok(bn, d)


def test_encoder_training_forward_backward_matches_aten():
    """Same weights, same input: outputs, input gradient, every parameter gradient and the BatchNorm running
    statistics of the HIP training path against the ATen one (causal window active: T' = 40 > 16)."""
    x, ln = synthetic_mel(3, 320, [320, 250, 199], 5)
    res = {}
    for native in (True, False):
        m = _train_model(CFG, 31, native)
        xin = x.cuda().requires_grad_()
        out, out_len, cache = m.encoder(xin, ln.cuda())
        assert cache is None and out_len.tolist() == [40, 32, 25]
        w = torch.linspace(-1, 1, out.numel(), device="cuda").view(out.shape)
        (out * w).sum().backward()
        res[native] = (out.detach(), xin.grad.detach(), {k: p.grad.detach() for k, p in m.encoder.named_parameters()},
                       {k: b.detach().clone() for k, b in m.encoder.named_buffers() if "running" in k})
    (o1, gx1, gp1, bn1), (o0, gx0, gp0, bn0) = res[True], res[False]
    _close(o1, o0, "encoder output")
    _close(gx1, gx0, "input gradient", 5e-4)
    assert gp1.keys() == gp0.keys() and len(gp1) > 50
    largest = max(g.abs().max().item() for g in gp0.values())
    for k in gp0:
        _close(gp1[k], gp0[k], f"gradient of {k}", 5e-4, floor=1e-3 * largest)
    for k in bn0:
        _close(bn1[k], bn0[k], f"buffer {k}")


def test_transducer_training_step_as_train_py_runs_it():
    """model(mels, mel_lengths, tokens, token_lengths) -> logits (B, T', U+1, V), backward through encoder,
    predictor and joint (train.py:302-336), HIP training path against ATen in fp32."""
    x, ln = synthetic_mel(2, 160, [160, 120], 6)
    tokens = torch.tensor([[3, 7, 1, 9], [5, 2, 0, 0]])
    tl = torch.tensor([4, 2])
    grads = {}
    for native in (True, False):
        m = _train_model(CFG, 32, native)
        logits, enc_len = m(x.cuda(), ln.cuda(), tokens.cuda(), tl.cuda())
        assert tuple(logits.shape) == (2, 20, 5, 29) and enc_len.tolist() == [20, 15]
        logits.float().log_softmax(-1)[..., 0].sum().backward()
        grads[native] = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    assert grads[True].keys() == grads[False].keys()
    largest = max(g.abs().max().item() for g in grads[False].values())
    for k in grads[False]:
        _close(grads[True][k], grads[False][k], f"gradient of {k}", 5e-4, floor=1e-3 * largest)


def _amp_grads(seed, native, dtype, x, ln):
    m = _train_model(CFG, seed, native)
    tokens = torch.tensor([[3, 7, 1, 9], [5, 2, 0, 0]]).cuda()
    tl = torch.tensor([4, 2]).cuda()
    if dtype is None:
        logits, _ = m(x.cuda(), ln.cuda(), tokens, tl)
    else:
        with torch.autocast("cuda", dtype=dtype):
            logits, _ = m(x.cuda(), ln.cuda(), tokens, tl)
    logits.float().log_softmax(-1)[..., 0].sum().backward()
    out = {k: p.grad.detach().double() for k, p in m.named_parameters() if p.grad is not None}
    for k, g in out.items():
        assert torch.isfinite(g).all(), k
    return out


def _rel_l2(got, want):
    assert got.keys() == want.keys()
    num = sum((got[k] - want[k]).pow(2).sum().item() for k in want)
    return (num / sum(v.pow(2).sum().item() for v in want.values())) ** 0.5


def test_amp_training_step_is_exact_products_of_bf16_rounded_operands(monkeypatch):
    """AMP as train.py:301 (autocast on the GPU): every Linear of the HIP path multiplies bf16-rounded operands and
    accumulates in fp32, forward and both backward products.  Checked against the same step with the library's GEMM
    replaced by an ATen fp32 product of the rounded operands: same function, so the gradients agree to accumulation
    order (the comparison with fp32 gradients below cannot pin this: a random-init model amplifies any bf16 rounding)."""
    x, ln = synthetic_mel(2, 160, [160, 120], 6)
    hip = _amp_grads(32, True, torch.bfloat16, x, ln)

    def emulated(a, w, bias, dt):
        with torch.autocast("cuda", enabled=False):
            out = a.to(dt).float() @ w.to(dt).float().t()
            return out if bias is None else out + bias
    monkeypatch.setattr(A, "_gemm_f32out", emulated)
    want = _amp_grads(32, True, torch.bfloat16, x, ln)
    # not 1e-6: a last-bit difference in an fp32 sum moves some later bf16 rounding, and the model amplifies that;
    # a different function (ATen's autocast, say) is 4e-2 or more away
    assert _rel_l2(hip, want) <= 1.5e-2, f"HIP AMP step against its ATen emulation: relative L2 {_rel_l2(hip, want):.3e}"


def test_amp_gradients_are_no_further_from_fp32_than_aten_autocast():
    """Against fp32 gradients, reduced-precision ones of a random-init model are off by anything between 4 % and
    100 % depending on the initialisation (the first block's attention has no pre-LayerNorm and saturated scores: one
    rounding flips a softmax row), whichever kernels compute them.  So the yardstick is ATen's own bf16 autocast over
    several initialisations: the median error of the HIP path must not exceed twice ATen's."""
    hip, aten = [], []
    for seed in range(40, 46):
        x, ln = synthetic_mel(2, 160, [160, 120], seed + 100)
        ref = _amp_grads(seed, False, None, x, ln)
        hip.append(_rel_l2(_amp_grads(seed, True, torch.bfloat16, x, ln), ref))
        aten.append(_rel_l2(_amp_grads(seed, False, torch.bfloat16, x, ln), ref))
    hip_med, aten_med = sorted(hip)[3], sorted(aten)[3]
    assert hip_med <= 2 * aten_med + 0.05, f"AMP gradient error, medians over 6 seeds: HIP {hip_med:.3e}, ATen bf16 {aten_med:.3e}"
    # fp16 autocast (train.py's dtype on CUDA) takes the same bf16 kernels
    x, ln = synthetic_mel(2, 160, [160, 120], 140)
    _amp_grads(40, True, torch.float16, x, ln)


def test_training_then_validation_uses_the_updated_weights():
    """An optimizer step on the HIP training path, then eval: the inference kernels see the new weights."""
    from oracle import fc_oracle as O
    m = _train_model(CFG, 33, True)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    x, ln = synthetic_mel(2, 160, [160, 120], 7)
    out, _, _ = m.encoder(x.cuda(), ln.cuda())
    out.pow(2).mean().backward()
    opt.step()
    m.eval()
    with torch.inference_mode():
        got, got_len, _ = m.encode(x.cuda(), ln.cuda())
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref, ref_len, _ = O.encoder_forward(sd, CFG, x, ln)
    assert torch.equal(got_len.cpu(), ref_len)
    assert (got.cpu() - ref).abs().max().item() <= 1e-4
