"""GPU parity: HIP kernels (through the C ABI) against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu

RTOL = 1e-3   # BASELINE.json north_star: within 1e-3 relative in fp32


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def ops():
    from style_transfer_amd import ops as o
    return o


@pytest.mark.parametrize("n,cin,cout,h,w,up,pad,relu", [
    (1, 8, 32, 8, 8, False, "replicate", True),
    (2, 3, 64, 37, 45, False, "replicate", True),      # Cin tail, odd sizes
    (1, 64, 64, 40, 33, False, "replicate", True),
    (2, 64, 128, 19, 23, False, "replicate", True),
    (1, 128, 256, 16, 16, False, "zero", True),        # loss-VGG style
    (1, 256, 256, 9, 11, True, "replicate", True),     # fused nearest x2 upsample
    (1, 512, 256, 5, 6, False, "replicate", True),
    (1, 64, 3, 40, 48, False, "replicate", False),     # last decoder conv: no ReLU, Cout=3
    (1, 20, 130, 7, 70, False, "zero", False),         # ragged channels
    (1, 16, 40, 1, 1, False, "replicate", True),       # degenerate 1x1 image
    # the 3-channel layers run on the vector-ALU kernels (csrc/conv3x3_small.hip): ragged tiles, both pads,
    # 1-pixel images, the upsampled / non-multiple-of-16 shapes that must fall back to the MFMA kernel
    (2, 3, 64, 17, 33, False, "zero", True),           # loss-VGG conv1_1: zero pad, tile edges in both directions
    (1, 3, 64, 1, 1, False, "replicate", True),
    (1, 3, 48, 9, 70, False, "replicate", False),      # cout % 16 == 0 but not 64
    (1, 3, 40, 9, 9, False, "zero", True),             # cout % 16 != 0 -> MFMA kernel
    (1, 3, 64, 6, 7, True, "replicate", True),         # upsample -> MFMA kernel
    (2, 64, 3, 17, 67, False, "zero", True),           # 64 -> 3 with zero pad and ReLU, ragged tiles
    (1, 128, 3, 33, 130, False, "replicate", False),   # 16 chunks, three column tiles
    (1, 8, 3, 1, 1, False, "replicate", False),
    (1, 12, 3, 5, 5, False, "replicate", False),       # cin % 8 != 0 -> MFMA kernel
    # w % 4 == 0, w >= 64 and >= 256 K pixels: the row-strip kernel (2-row strips; 8-row strips from 2 M pixels) with
    # 16 / 32 / 64 lanes per row, a ragged last column tile, strips that overhang the image, both pads
    (4, 8, 3, 964, 68, False, "zero", True),
    (2, 8, 3, 1000, 132, False, "replicate", False),
    (4, 16, 3, 255, 260, False, "replicate", True),
    (1, 8, 3, 513, 512, False, "zero", False),
    (8, 8, 3, 4001, 68, False, "replicate", False),
    (1, 8, 3, 1030, 2048, False, "zero", True),
    (3, 24, 3, 1021, 700, False, "replicate", False),
    (1, 512, 3, 512, 512, False, "replicate", False),  # 72 KB weight image in LDS (dynamic-LDS opt-in), strips
    (1, 512, 3, 9, 20, False, "zero", True),           # ... and the LDS-tiled kernel
])
def test_conv3x3(ops, n, cin, cout, h, w, up, pad, relu):
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g)
    xin = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest") if up else x
    if pad == "replicate":
        ref = torch.nn.functional.conv2d(R.symmetric_pad(xin), wt, b)
    else:
        ref = torch.nn.functional.conv2d(xin, wt, b, padding=1)
    if relu:
        ref = torch.relu(ref)
    wp = ops.pack_conv3x3_weights(wt.cuda())
    y = ops.conv3x3(x.cuda(), wp, b.cuda(), cout, upsample2x=up, pad_mode=pad, relu=relu)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 1e-5


def test_conv1x1_and_pool(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 17, 19, generator=g)
    w = torch.randn(3, 3, 1, 1, generator=g) * 50
    b = torch.randn(3, generator=g) * 100
    y = ops.conv1x1_3to3(x.cuda(), w.cuda(), b.cuda())
    assert rel_err(y, torch.nn.functional.conv2d(x, w, b)) < 1e-6
    z = torch.randn(2, 5, 17, 19, generator=g)
    for ceil in (True, False):
        p = ops.maxpool2x2(z.cuda(), ceil)
        ref = torch.nn.functional.max_pool2d(z, 2, 2, 0, ceil_mode=ceil)
        assert p.shape == ref.shape and torch.equal(p.cpu(), ref)


@pytest.mark.parametrize("hc,wc,hs,ws", [(7, 9, 5, 4), (64, 64, 64, 64), (64, 64, 32, 48), (128, 128, 128, 128),
                                          (160, 132, 24, 20)])
@pytest.mark.parametrize("alpha", [1.0, 0.5, 1.5])
def test_adain(ops, hc, wc, hs, ws, alpha):
    g = torch.Generator().manual_seed(hc * 7 + ws)
    c = torch.randn(2, 6, hc, wc, generator=g) * 2.0 + 0.5
    s = torch.randn(2, 6, hs, ws, generator=g) * 0.7 - 1.0
    eps_c, eps_s = 0.02, 1e-5
    ref = R.blend(R.adain(c, s, style_eps=eps_s, content_eps=eps_c), c, alpha)
    out, stats = ops.adain(c.cuda(), s.cuda(), eps_c, eps_s, alpha=alpha, save_stats=True)
    assert rel_err(out, ref) < 1e-5
    mc, sc = R.calc_mean_std(c, eps_c)
    ms, ss = R.calc_mean_std(s, eps_s)
    st = stats.cpu().view(2, 6, 4)
    np.testing.assert_allclose(st[..., 0], mc.view(2, 6), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st[..., 1], sc.view(2, 6), rtol=1e-5)
    np.testing.assert_allclose(st[..., 2], ms.view(2, 6), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st[..., 3], ss.view(2, 6), rtol=1e-5)


def test_adain_device_eps_matches_host_eps(ops):
    """st_adain_fwd_dev_eps (eps read from device memory) == st_adain_fwd (eps by value), bit for bit, for the
    three kernel variants (register, 64-KB planes, 256-KB planes) and the statistics it saves."""
    g = torch.Generator().manual_seed(3)
    for (n, c, hc, hs) in ((2, 16, 24, 17), (1, 8, 128, 100), (1, 4, 256, 256)):
        content = torch.randn(n, c, hc, hc, generator=g).cuda()
        style = (torch.randn(n, c, hs, hs, generator=g) * 2 + 0.5).cuda()
        eps = torch.tensor([3.1e-3, 7.7e-2], device="cuda")
        a, sa = ops.adain(content, style, 3.1e-3, 7.7e-2, 0.7, save_stats=True)
        b, sb = ops.adain(content, style, 0.0, 0.0, 0.7, save_stats=True, eps_dev=eps)
        torch.cuda.synchronize()
        assert torch.equal(a, b) and torch.equal(sa, sb)


@pytest.mark.parametrize("shape,offset", [((2, 8, 256, 256), 0.0), ((1, 4, 512, 512), 0.0), ((1, 6, 256, 128), 3.0),
                                          ((2, 3, 192, 172), -1.5)])
def test_mean_std_large_planes(ops, shape, offset):
    """Planes of 128 KB and more take the one-pass kernel (shifted sums, csrc/adain.hip meanstd_big_kernel): mean and
    unbiased std + eps against float64 (transformer.py:15-24), on post-ReLU-like data with an offset and an outlier
    at the first element."""
    g = torch.Generator().manual_seed(shape[2])
    x = torch.relu(torch.randn(*shape, generator=g)) * 0.8 + offset
    x[:, :, 0, 0] = 50.0                                   # an outlier: the shift K is a 64-sample average
    mean, std = ops.mean_std(x.cuda(), 1e-5)
    xd = x.double().flatten(2)
    ref_mean = xd.mean(dim=2).view(shape[0], shape[1], 1, 1)
    ref_std = (xd.var(dim=2) + 1e-5).sqrt().view(shape[0], shape[1], 1, 1)
    assert rel_err(mean, ref_mean) < 2e-6
    assert rel_err(std, ref_std) < 1e-5


def test_adain_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "adain.npz"))
    c, s = torch.from_numpy(g["content"]), torch.from_numpy(g["style"])
    # eps_limit 0.1: content_eps -> sqrt(4e-4)=0.02, style_eps -> sqrt(2.5e-3)=0.05 (reference output)
    out, _ = ops.adain(c.cuda(), s.cuda(), float(g["eps_0.1"][1]), float(g["eps_0.1"][0]))
    assert rel_err(out, torch.from_numpy(g["out_0.1"])) < 1e-5
    mean, std = ops.mean_std(c.cuda(), 1e-5)
    assert rel_err(mean, torch.from_numpy(g["mean"])) < 1e-5
    assert rel_err(std, torch.from_numpy(g["std"])) < 1e-5


@pytest.mark.parametrize("compute", ["f32", "bf16"])
@pytest.mark.parametrize("n,cin,cout,h,w", [
    (2, 64, 64, 37, 45),      # odd sizes: ceil-mode windows of one row / one column at the edges
    (1, 128, 128, 16, 64),
    (1, 256, 256, 9, 33),
    (1, 3, 64, 5, 1),         # single column
    (1, 16, 20, 66, 70),      # ragged channels, several tiles
])
def test_conv3x3_fused_ceil_pool(ops, compute, n, cin, cout, h, w):
    """[pad, conv, ReLU, MaxPool2d(ceil_mode=True)] in one launch (generator.py:51-53) vs conv then pool."""
    g = torch.Generator().manual_seed(cin + cout + h + w)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g)
    with R.conv_compute(compute):
        ref = torch.nn.functional.max_pool2d(torch.relu(R.conv3x3_padded(R.symmetric_pad(x), wt, b)), 2, 2, 0,
                                             ceil_mode=True)
    wp = ops.pack_conv3x3_weights(wt.cuda(), compute=compute)
    y = ops.conv3x3_pool(x.cuda(), wp, b.cuda(), cout, pad_mode="replicate", relu=True)
    y2 = ops.maxpool2x2(ops.conv3x3(x.cuda(), wp, b.cuda(), cout, pad_mode="replicate", relu=True), True)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    if cin == 3:
        # the unfused launch of a 3-channel layer runs on its own K = 27 kernel (csrc/conv3x3_small.hip for fp32,
        # csrc/conv3x3_bf16_cin3.hip for bf16; no network has a pool behind its 3-channel layer): same products,
        # different fp32 summation order
        assert rel_err(y, y2) < 1e-6
    else:
        assert torch.equal(y, y2)                   # bit-identical to the unfused pair of launches
    assert rel_err(y, ref) < (1e-5 if compute == "f32" else 1e-4)
