"""GPU parity of the "bf16x3" compute mode (csrc/conv3x3_bf16.hip, SPLIT = true): fp32 operands split
exactly into hi + lo bf16 halves, products evaluated as hi*hi + hi*lo + lo*hi on the bf16 matrix pipe.

The bar is the fp32 bar of the task (1e-3 relative against the reference's fp32 numerics, i.e. the plain
fp32 oracle -- no rounding emulation), and the tests hold the mode to a much tighter one: 1e-4 max-norm
relative per layer and for the whole 19-conv stylise pipeline (measured: ~1e-5).
"""
import pytest
import torch

from oracle import ref_ops as R
from test_gpu_bf16 import CASES, _ref_conv, rel_err, rel_l2

pytestmark = pytest.mark.gpu

X3_TOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    from style_transfer_amd import ops as o
    return o


@pytest.mark.parametrize("n,cin,cout,h,w,up,pad,relu", CASES)
def test_conv3x3_bf16x3_forward(ops, n, cin, cout, h, w, up, pad, relu):
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = _ref_conv(x, wt, b, up, pad, relu)
    wp = ops.pack_conv3x3_weights(wt.cuda(), compute="bf16x3")
    assert ops.conv_family(wp, cout, cin) == "bf16x3"
    y = ops.conv3x3(x.cuda(), wp, b.cuda(), cout, upsample2x=up, pad_mode=pad, relu=relu)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert rel_err(y, ref) < X3_TOL


@pytest.mark.parametrize("n,cin,cout,h,w,pad", [(2, 3, 64, 37, 45, "replicate"), (1, 64, 128, 40, 33, "replicate"),
                                                (1, 128, 256, 16, 16, "zero")])
def test_conv3x3_bf16x3_pool(ops, n, cin, cout, h, w, pad):
    g = torch.Generator().manual_seed(5 + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = torch.nn.functional.max_pool2d(_ref_conv(x, wt, b, False, pad, True), 2, 2, ceil_mode=True)
    wp = ops.pack_conv3x3_weights(wt.cuda(), compute="bf16x3")
    y = ops.conv3x3_pool(x.cuda(), wp, b.cuda(), cout, pad_mode=pad, relu=True)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert rel_err(y, ref) < X3_TOL


@pytest.mark.parametrize("n,cin,cout,h,w,up,pad,relu", [c for c in CASES if not (c[5] and c[6] == "zero")])
def test_conv3x3_bf16x3_dgrad(ops, n, cin, cout, h, w, up, pad, relu):
    g = torch.Generator().manual_seed(7 + n * 1000 + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g)
    y_ref = _ref_conv(x, wt, b, up, pad, relu)
    dy = torch.randn(y_ref.shape, generator=g)
    (dx_ref,) = torch.autograd.grad(y_ref, x, dy)
    wpt = ops.pack_conv3x3_weights(wt.cuda(), transpose_flip=True, compute="bf16x3")
    # gate with the ORACLE's activation so that a pre-activation within 1e-5 of zero cannot flip a gate
    dx = ops.conv3x3_dgrad(dy.cuda(), y_ref.detach().cuda() if relu else None, wpt, cin, up, pad)
    torch.cuda.synchronize()
    assert dx.shape == dx_ref.shape
    assert rel_err(dx, dx_ref) < X3_TOL


def test_stylize_bf16x3_pipeline():
    from style_transfer_amd import precision
    from style_transfer_amd.generator import Decoder, Encoder
    from style_transfer_amd.pipeline import stylize
    from style_transfer_amd.transformer import AdaIN
    enc_sd, dec_sd = R.make_encoder_state(0), R.make_decoder_state(1)
    enc = Encoder(weights=enc_sd).cuda().eval()
    dec = Decoder().cuda().eval()
    dec.load_state_dict(dec_sd)
    ada = AdaIN().cuda()
    g = torch.Generator().manual_seed(11)
    c = torch.rand(2, 3, 72, 88, generator=g)
    s = torch.rand(2, 3, 64, 64, generator=g)
    with precision.compute_dtype("bf16x3"):
        out = stylize(enc, dec, ada, c.cuda(), s.cuda(), alpha=1.0)
    out32 = stylize(enc, dec, ada, c.cuda(), s.cuda(), alpha=1.0)        # same modules, fp32 mode again
    torch.cuda.synchronize()
    ref32 = R.stylize(c, s, enc_sd, dec_sd, alpha=1.0)
    assert rel_err(out32, ref32) < 1e-3
    print("bf16x3 pipeline max-norm rel err", rel_err(out, ref32), "rel L2", rel_l2(out, ref32),
          "| fp32 MFMA path", rel_err(out32, ref32))
    assert rel_err(out, ref32) < X3_TOL          # 10x inside the task's fp32 tolerance (1e-3)


def test_train_step_bf16x3_matches_fp32():
    """Training step in bf16x3 mode vs the exact-fp32 HIP path: every loss within 1e-3 relative, decoder
    gradients of the content / style terms within 1e-2 (variation: 5e-2) relative L2 per parameter (forward and data
    gradient on the split kernels, weight gradient on the fp32 kernel; isolated ReLU gates may flip where a
    pre-activation is within 1e-5 of zero).  The histogram term's gradient is left out for the reason given in
    tests/test_gpu_bf16.py (discontinuous clamp mask)."""
    from style_transfer_amd import precision
    from test_gpu_train import _build
    step, *_ = _build(False, [21], [2, 7, 14, 19, 25], 0.1)
    g = torch.Generator().manual_seed(5)
    content = torch.rand(2, 3, 64, 64, generator=g).cuda()
    style = torch.rand(2, 3, 64, 64, generator=g).cuda()
    params = list(step.decoder.named_parameters())

    def run(i):
        for _, p in params + list(step.transformer.named_parameters()):
            p.grad = None
        losses = step.forward_losses(content, style)
        losses[i].backward()
        return [float(l.detach()) for l in losses], [p.grad.clone() for _, p in params]

    for i, name in enumerate(["content", "style", "variation"]):
        l32, g32 = run(i)
        with precision.compute_dtype("bf16x3"):
            lx3, gx3 = run(i)
        for a, b in zip(lx3, l32):
            assert abs(a - b) <= 1e-3 * max(abs(b), 1e-3), (name, a, b)
        for (pname, _), a, b in zip(params, gx3, g32):
            if name == "variation" and pname.endswith("28.bias"):
                continue                                     # analytically zero
            # TVLoss' gradient is sign(difference) (losses.py:104-116): a 1e-5 perturbation of the decoded image
            # flips the sign wherever neighbouring pixels are that close, so that term gets a wider bar
            assert rel_l2(a, b) < (5e-2 if name == "variation" else 1e-2), (name, pname, rel_l2(a, b))
