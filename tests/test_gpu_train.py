"""Training path (train.py:302-336: model.forward + loss.backward) on the HIP kernels of conformer_amd.autograd,
against ATen: each autograd function alone (outputs and every gradient), then the whole encoder and the whole
transducer forward in training mode with `native_training` on and off from the same weights -- outputs, gradients
of every parameter and of the input, and the BatchNorm running statistics left behind.  fp32 throughout (the
training default; the reference's fp32 gate of 1e-4 is applied relative to each tensor's largest element),
then AMP: bf16 operands with fp32 accumulation against the fp32 gradients."""
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
    predictor and joint (train.py:302-336), HIP training path against ATen; then the same step under autocast."""
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
    # AMP as train.py:301 (autocast on the GPU): bf16 operands, fp32 accumulation, fp32 gradients
    m = _train_model(CFG, 32, True)
    with torch.autocast("cuda", dtype=torch.float16):
        logits, _ = m(x.cuda(), ln.cuda(), tokens.cuda(), tl.cuda())
    logits.float().log_softmax(-1)[..., 0].sum().backward()
    num = den = 0.0
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), k
            num += (p.grad.float() - grads[False][k]).pow(2).sum().item()
            den += grads[False][k].pow(2).sum().item()
    assert (num / den) ** 0.5 <= 5e-2, f"AMP gradients: relative L2 {(num / den) ** 0.5:.3e}"


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
