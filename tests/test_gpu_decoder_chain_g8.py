"""GPU parity of the decoder's training pass in bf16 mode with G8 hand-overs in both directions
(autograd_conv.DecoderChainG8Function: st_conv3x3_dgrad_bf16_g8_chain, st_conv3x3_wgrad_bf16_pre) against the
fp32-activation chain (DecoderChainFunction, pinned to the same-rounding oracle by tests/test_gpu_bf16.py).

Output, input gradient and weight gradients must be BIT-IDENTICAL (a G8 / bf16 hand-over is the rounding the
consumer's staging applies to the fp32 form anyway; fold, gate and GEMMs do the same fp32 arithmetic); the bias
gradients are summed in another fixed order and agree to fp32 rounding.  One stated exception: with the G8 hand-over
to the 3-channel top layer (TOP_G8, round 3) that layer's vector-ALU weight gradient sees the bf16 activation.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _decoder(seed=1):
    from oracle import ref_ops as R
    from style_transfer_amd.generator import Decoder
    dec = Decoder(do_train=False).cuda()
    dec.load_state_dict(R.make_decoder_state(seed))
    return dec


def _run(dec, x0, gy, chain):
    from style_transfer_amd import precision
    dec.G8_CHAIN = chain
    for p in dec.parameters():
        p.grad = None
    x = x0.clone().requires_grad_(True)
    with precision.compute_dtype("bf16"):
        y = dec(x)
        y.backward(gy)
    return y.detach(), x.grad, [p.grad.clone() for p in dec.parameters()]


@pytest.mark.parametrize("n,cin,h,w,up,pad", [(2, 64, 32, 64, False, "replicate"), (1, 64, 19, 37, False, "replicate"),
                                               (2, 16, 8, 24, True, "replicate"), (1, 8, 5, 5, False, "zero")])
def test_top_layer_weight_gradient_from_a_g8_input(n, cin, h, w, up, pad):
    """st_conv3x3_wgrad_cout3_xg8 reads the G8 activation and rounds dy to bf16 for the weight gradient (the bf16
    mode multiplies bf16 operands; oracle.ref_ops._ConvBf16Operands.backward) while the bias gradient sums dy as it
    is; the fp32 entry point on the same values does the same fp32 arithmetic in the same order: bit-identical."""
    from style_transfer_amd import ops
    g = torch.Generator().manual_seed(n + cin + h)
    x = torch.randn(n, cin, h, w, generator=g).cuda()
    s = 2 if up else 1
    dy = torch.randn(n, 3, h * s, w * s, generator=g).cuda()
    xg = ops.to_g8(x)
    dw, db = ops.conv3x3_wgrad_cout3_g8(xg, dy, upsample2x=up, pad_mode=pad)
    x32 = ops.from_g8(xg)
    dw0, _ = ops.conv3x3_wgrad(x32, dy.to(torch.bfloat16).float(), None, want_bias=True, upsample2x=up, pad_mode=pad,
                               compute="bf16")
    _, db0 = ops.conv3x3_wgrad(x32, dy, None, want_bias=True, upsample2x=up, pad_mode=pad, compute="bf16")
    assert float(dw0.abs().max()) > 0
    assert torch.equal(dw, dw0) and torch.equal(db, db0)


@pytest.mark.parametrize("top_g8", [False, True])
@pytest.mark.parametrize("n,h,w", [(2, 4, 32), (1, 8, 64), (2, 32, 32)])
def test_decoder_g8_chain_matches_the_fp32_activation_chain(n, h, w, top_g8, monkeypatch):
    from style_transfer_amd.autograd_conv import DecoderChainG8Function
    monkeypatch.setattr(DecoderChainG8Function, "TOP_G8", top_g8)
    dec = _decoder()
    g = torch.Generator().manual_seed(3)
    x0 = torch.relu(torch.randn(n, 512, h, w, generator=g)).cuda()
    gy = torch.randn(n, 3, 8 * h, 8 * w, generator=g).cuda()
    seq = dec.decoder
    specs, up = [], False
    for idx in dec.CONV3:
        specs.append((idx, int(seq[idx].weight.shape[0]), up, idx != 28))
        up = idx in dec.UP_AFTER
    params = [t for idx in dec.CONV3 for t in (seq[idx].weight, seq[idx].bias)]
    assert DecoderChainG8Function.usable(x0, specs, params)
    try:
        y0, dx0, gr0 = _run(dec, x0, gy, False)
        y1, dx1, gr1 = _run(dec, x0, gy, True)
    finally:
        dec.G8_CHAIN = True
    assert torch.equal(y1, y0)
    assert float(dx0.abs().max()) > 0
    assert torch.equal(dx1, dx0)
    for k, (a, b) in enumerate(zip(gr1, gr0)):
        if a.dim() == 4 and top_g8 and k == len(gr1) - 2:
            # the 64 -> 3 layer's weight gradient multiplies dy with the bf16 values its forward multiplied (the G8
            # input) where the fp32-activation chain multiplies with the unrounded fp32 activation: bf16 rounding of
            # one operand, <= 2^-9 per product; with a random-sign dy the sum cancels, so relative to the SUM it is
            # more (measured 1.5e-3 ... 5.6e-3 on these shapes).  The bit-level check of the kernel is
            # test_top_layer_weight_gradient_from_a_g8_input.
            assert float((a - b).norm() / b.norm()) < 2e-2, f"top-layer weight gradient {float((a - b).norm() / b.norm())}"
        elif a.dim() == 4:
            assert torch.equal(a, b), f"weight gradient {k}"
        else:
            torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5 * float(b.abs().max()))


def test_shapes_outside_the_staged_form_take_the_other_chain():
    from style_transfer_amd.autograd_conv import DecoderChainG8Function
    dec = _decoder()
    x = torch.rand(1, 512, 5, 7).cuda()
    seq = dec.decoder
    specs, up = [], False
    for idx in dec.CONV3:
        specs.append((idx, int(seq[idx].weight.shape[0]), up, idx != 28))
        up = idx in dec.UP_AFTER
    params = [t for idx in dec.CONV3 for t in (seq[idx].weight, seq[idx].bias)]
    assert not DecoderChainG8Function.usable(x, specs, params)
    gy = torch.randn(1, 3, 40, 56).cuda()
    y, dx, _ = _run(dec, x, gy, True)          # falls back without error
    assert y.shape == (1, 3, 40, 56) and dx.shape == x.shape


def test_frozen_decoder_only_the_input_gradient():
    dec = _decoder()
    for p in dec.parameters():
        p.requires_grad_(False)
    from style_transfer_amd import precision
    g = torch.Generator().manual_seed(4)
    x0 = torch.relu(torch.randn(1, 512, 8, 32, generator=g)).cuda()
    gy = torch.randn(1, 3, 64, 256, generator=g).cuda()
    outs = []
    for chain in (False, True):
        dec.G8_CHAIN = chain
        x = x0.clone().requires_grad_(True)
        with precision.compute_dtype("bf16"):
            dec(x).backward(gy)
        outs.append(x.grad)
    dec.G8_CHAIN = True
    assert torch.equal(outs[0], outs[1])
