"""GPU parity of the backward kernels (conv dgrad / wgrad, pool) against torch autograd on the CPU oracle ops."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def ref_layer(x, w, b, up, pad, relu):
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    y = F.conv2d(R.symmetric_pad(xin), w, b) if pad == "replicate" else F.conv2d(xin, w, b, padding=1)
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("n,cin,cout,h,w,up,pad,relu", [
    (1, 8, 32, 8, 8, False, "replicate", True),
    (2, 64, 64, 13, 37, False, "replicate", True),
    (2, 64, 128, 19, 23, False, "zero", True),
    (1, 256, 128, 6, 5, True, "replicate", True),      # fused upsample adjoint
    (1, 64, 3, 40, 48, False, "replicate", False),     # last decoder conv
    (1, 512, 256, 3, 5, False, "replicate", True),
    (2, 20, 130, 7, 70, False, "zero", False),         # ragged channels
    (1, 32, 40, 1, 1, False, "replicate", True),       # 1x1 image: all nine taps fold onto one pixel
    (1, 16, 16, 2, 64, True, "replicate", True),
])
def test_conv_backward(n, cin, cout, h, w, up, pad, relu):
    from style_transfer_amd import ops
    g = torch.Generator().manual_seed(cin * 31 + cout + h * 7 + w)
    x = torch.randn(n, cin, h, w, generator=g).requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True)
    y = ref_layer(x, wt, b, up, pad, relu)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)

    xd, wd, bd, dyd = x.detach().cuda(), wt.detach().cuda(), b.detach().cuda(), dy.cuda()
    yd = ops.conv3x3(xd, ops.pack_conv3x3_weights(wd), bd, cout, upsample2x=up, pad_mode=pad, relu=relu)
    gate = yd if relu else None
    dx = ops.conv3x3_dgrad(dyd, gate, ops.pack_conv3x3_weights(wd, transpose_flip=True), cin, up, pad)
    dw, db = ops.conv3x3_wgrad(xd, dyd, gate, True, up, pad)
    torch.cuda.synchronize()
    assert dx.shape == x.shape and dw.shape == wt.shape and db.shape == b.shape
    assert rel_err(dx, x.grad) < 2e-5
    assert rel_err(dw, wt.grad) < 2e-5
    assert rel_err(db, b.grad) < 2e-5


@pytest.mark.parametrize("h,w", [(8, 8), (7, 9), (1, 5), (16, 3)])
@pytest.mark.parametrize("ceil", [True, False])
def test_maxpool_backward(h, w, ceil):
    from style_transfer_amd import ops
    if not ceil and (h < 2 or w < 2):
        pytest.skip("empty floor-mode output")
    g = torch.Generator().manual_seed(h * 10 + w)
    x = torch.randn(2, 3, h, w, generator=g)
    x[0, 0, 0, :2] = 1.5                     # a tie inside a window: first max wins
    x = x.requires_grad_(True)
    y = F.max_pool2d(x, 2, 2, 0, ceil_mode=ceil)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dx = ops.maxpool2x2_bwd(x.detach().cuda(), dy.cuda(), ceil)
    assert torch.equal(dx.cpu(), x.grad)


def test_decoder_autograd_golden(golden_dir):
    """Decoder module: dgrad + wgrad through all nine fused layers vs the reference's autograd."""
    from style_transfer_amd.generator import Decoder
    g = np.load(os.path.join(golden_dir, "generator.npz"))
    dec = Decoder(do_train=True).cuda()
    dec.load_state_dict(R.make_decoder_state(1))
    fin = torch.randn(1, 512, 3, 5, generator=torch.Generator().manual_seed(12)).cuda().requires_grad_(True)
    yo = dec(fin)
    gout = torch.randn(yo.shape, generator=torch.Generator().manual_seed(13)).cuda()
    (yo * gout).sum().backward()
    assert rel_err(yo, torch.from_numpy(g["g_out"])) < 1e-3
    assert rel_err(fin.grad, torch.from_numpy(g["g_din"])) < 1e-3
    assert rel_err(dec.decoder[28].weight.grad, torch.from_numpy(g["g_dw28"])) < 1e-3
    assert rel_err(dec.decoder[28].bias.grad, torch.from_numpy(g["g_db28"])) < 1e-3
    assert rel_err(dec.decoder[1].weight.grad[:8, :16], torch.from_numpy(g["g_dw1_slice"])) < 1e-3
    assert rel_err(dec.decoder[1].bias.grad, torch.from_numpy(g["g_db1"])) < 1e-3
    assert rel_err(dec.decoder[18].weight.grad[:8, :16], torch.from_numpy(g["g_dw18_slice"])) < 1e-3
