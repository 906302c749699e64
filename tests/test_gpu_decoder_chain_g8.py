"""GPU parity of the decoder's training pass in bf16 mode with G8 hand-overs in both directions
(autograd_conv.DecoderChainG8Function: st_conv3x3_dgrad_bf16_g8_chain, st_conv3x3_wgrad_bf16_pre) against the
fp32-activation chain (DecoderChainFunction, pinned to the same-rounding oracle by tests/test_gpu_bf16.py).

Output, input gradient and weight gradients must be BIT-IDENTICAL (a G8 / bf16 hand-over is the rounding the
consumer's staging applies to the fp32 form anyway; fold, gate and GEMMs do the same fp32 arithmetic); the bias
gradients are summed in another fixed order and agree to fp32 rounding.
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


@pytest.mark.parametrize("n,h,w", [(2, 4, 32), (1, 8, 64), (2, 32, 32)])
def test_decoder_g8_chain_matches_the_fp32_activation_chain(n, h, w):
    from style_transfer_amd.autograd_conv import DecoderChainG8Function
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
        if a.dim() == 4:
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
