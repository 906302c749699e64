"""GPU parity of the bf16 layer-to-layer form (st_conv3x3_fwd_bf16_g8, csrc/conv3x3_bf16.hip): activations handed
from conv to conv in the channel-group layout G8 = bf16 [n][c/8][h][w][8].

The bar is BIT-IDENTITY with the fp32-activation bf16 kernels (st_conv3x3_fwd_bf16 / st_conv3x3_pool_fwd_bf16, which
tests/test_gpu_bf16.py pins to the oracle): a G8 value is the round-to-nearest-even those kernels apply to their
input anyway, the chunk/tap accumulation order is the same, and rounding commutes with ReLU and max-pool.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from style_transfer_amd import ops as o
    return o


def _layer(ops, n, cin, cout, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g).cuda()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    return x, ops.pack_conv3x3_weights(wt, False, "bf16"), b


CASES = [
    # n, cin, cout, h, w, up, relu, pool
    (2, 64, 64, 40, 40, False, True, False),     # 64-ch tile (32 rows x 32)
    (1, 64, 128, 33, 47, False, True, False),    # ragged both ways, 128-ch tile
    (2, 128, 128, 37, 45, False, True, True),    # fused ceil-mode pool, odd sizes
    (1, 64, 64, 64, 96, False, True, True),      # pool on the 64-ch tile
    (1, 256, 256, 16, 16, True, True, False),    # fused x2 upsample
    (1, 256, 128, 9, 13, True, True, False),     # upsample, ragged
    (1, 512, 256, 8, 8, False, True, False),     # tile larger than the image
    (1, 128, 64, 48, 32, False, False, False),   # no ReLU
    (3, 16, 8, 20, 20, False, True, False),      # one chunk, one channel group
    (1, 256, 512, 24, 40, False, True, False),   # four channel tiles per pixel tile
    # more than 256 tiles: the G8 -> G8 bf16 kernel is persistent (256 blocks walk the tiles, the next tile's first
    # chunk is prefetched under the epilogue)
    (3, 64, 64, 320, 320, False, True, False),   # 300 tiles of 32 rows, uneven tail
    (3, 64, 64, 320, 320, False, True, True),    # ... with the fused pool
    (3, 64, 256, 160, 192, False, True, False),  # 360 tiles, two channel tiles, plain tile order (180 % 8 != 0)
    (4, 32, 256, 160, 192, False, True, False),  # 480 tiles, XCD-swizzled tile order
    (2, 128, 128, 120, 100, True, True, False),  # upsample to 240 x 200: 2 x 15 x 7 = 210 ... x 1 channel tile
    (5, 16, 128, 150, 100, True, True, False),   # upsample to 300 x 200: 5 x 19 x 7 = 665 tiles
]


@pytest.mark.parametrize("n,cin,cout,h,w,up,relu,pool", CASES)
def test_g8_in_g8_out_is_bit_identical_to_fp32_activation_kernels(ops, n, cin, cout, h, w, up, relu, pool):
    x, wp, b = _layer(ops, n, cin, cout, h, w, seed=cin + cout + h)
    xg = ops.to_g8(x)
    xq = ops.from_g8(xg)                      # fp32 tensor holding the bf16 values: the reference kernels' view
    if pool:
        ref = ops.conv3x3_pool(xq, wp, b, cout, "replicate", relu)
    else:
        ref = ops.conv3x3(xq, wp, b, cout, up, "replicate", relu)
    got = ops.conv3x3_g8(xg, wp, b, cout, up, "replicate", relu, pool, y_g8=True)
    assert ops.is_g8(got) and tuple(got.shape) == (n, cout // 8) + tuple(ref.shape[2:]) + (8,)
    assert torch.equal(got, ops.to_g8(ref))
    got32 = ops.conv3x3_g8(xg, wp, b, cout, up, "replicate", relu, pool, y_g8=False)
    assert torch.equal(got32, ref)


@pytest.mark.parametrize("compute", ["bf16", "bf16x3"])
@pytest.mark.parametrize("n,cin,cout,h,w,up,relu,pool", [c for c in CASES if not c[5]])
def test_g8_zero_padding_is_bit_identical(ops, compute, n, cin, cout, h, w, up, relu, pool):
    """Zero padding (the loss VGG's convs, trainer.py:165-199) with a G8 input: LDS-DMA copies the clamped pixel and
    the copying lane zeroes its padding positions afterwards; same bits as the fp32-activation kernels."""
    g = torch.Generator().manual_seed(cin + 7 * cout + h)
    x = torch.randn(n, cin, h, w, generator=g).cuda() + 0.5         # non-zero border values
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    wp = ops.pack_conv3x3_weights(wt, False, compute)
    if compute == "bf16":
        xg = ops.to_g8(x)
        xq = ops.from_g8(xg)
    else:
        xg, xq = ops.to_g8_pair(x), x
    ref = ops.conv3x3_pool(xq, wp, b, cout, "zero", relu) if pool else ops.conv3x3(xq, wp, b, cout, False, "zero", relu)
    got32 = ops.conv3x3_g8(xg, wp, b, cout, False, "zero", relu, pool, y_g8=False)
    assert torch.equal(got32, ref)
    got = ops.conv3x3_g8(xg, wp, b, cout, False, "zero", relu, pool, y_g8=True)
    assert torch.equal(got, ops.to_g8(ref) if compute == "bf16" else ops.to_g8_pair(ref))


@pytest.mark.parametrize("n,cin,cout,h,w,up,pool", [
    (2, 3, 64, 37, 45, False, False),      # first encoder layer (channel tail) -> G8
    (1, 512, 256, 16, 24, False, False),   # first decoder layer: fp32 AdaIN output -> G8
    (1, 64, 64, 32, 64, False, True),
    (1, 128, 128, 8, 8, True, False),
])
def test_fp32_in_g8_out(ops, n, cin, cout, h, w, up, pool):
    x, wp, b = _layer(ops, n, cin, cout, h, w, seed=7 + cin)
    ref = ops.conv3x3_pool(x, wp, b, cout, "replicate", True) if pool else ops.conv3x3(x, wp, b, cout, up, "replicate", True)
    got = ops.conv3x3_g8(x, wp, b, cout, up, "replicate", True, pool, y_g8=True)
    assert torch.equal(got, ops.to_g8(ref))


def test_g8_in_three_channel_out(ops):
    """Last decoder layer: G8 input, 3 output channels, fp32 NCHW result, no ReLU."""
    x, wp, b = _layer(ops, 2, 64, 3, 40, 56, seed=5)
    xg = ops.to_g8(x)
    ref = ops.conv3x3(ops.from_g8(xg), wp, b, 3, False, "replicate", False)
    got = ops.conv3x3_g8(xg, wp, b, 3, False, "replicate", False, False, y_g8=False)
    assert torch.equal(got, ref)


def test_g8_argument_checks(ops):
    from style_transfer_amd._lib import StHipError
    x, wp, b = _layer(ops, 1, 16, 12, 8, 8, seed=1)
    with pytest.raises(StHipError):            # G8 output needs cout % 8 == 0
        ops.conv3x3_g8(ops.to_g8(x), wp, b, 12, False, "replicate", True, False, y_g8=True)
    x, wp, b = _layer(ops, 1, 16, 16, 8, 8, seed=1)
    with pytest.raises(StHipError):            # pool and upsample exclude each other
        ops.conv3x3_g8(ops.to_g8(x), wp, b, 16, True, "replicate", True, True, y_g8=True)


@pytest.mark.parametrize("compute,shape", [("bf16", (2, 3, 75, 91)), ("bf16x3", (2, 3, 75, 91)),
                                           ("bf16", (16, 3, 512, 512))])     # BASELINE configs[1] size
def test_encoder_decoder_g8_chain_is_bit_identical_to_layerwise_bf16(compute, shape):
    """Encoder / Decoder forward in bf16 / bf16x3 mode (G8 hand-overs between the nine convs of each) against the
    same layers run one by one through the fp32-activation kernels of that mode; odd sizes so that the ceil-mode
    pools and ragged tiles are on the path."""
    from oracle import ref_ops as R
    from style_transfer_amd import ops, precision
    from style_transfer_amd.generator import Decoder, Encoder
    enc = Encoder(weights=R.make_encoder_state(0)).cuda().eval()
    dec = Decoder().cuda().eval()
    dec.load_state_dict(R.make_decoder_state(1))
    g = torch.Generator().manual_seed(3)
    x = torch.rand(*shape, generator=g).cuda()
    with torch.no_grad(), precision.compute_dtype(compute):
        feat = enc(x)
        img = dec(feat)

    def pack(w):
        small = compute == "bf16x3" and 3 in (int(w.shape[0]), int(w.shape[1]))   # exact fp32 vector-ALU kernels
        return ops.pack_conv3x3_weights(w, False, "f32" if small else compute)
    # layer by layer, fp32 activations in HBM
    seq = enc.encode
    y = ops.conv1x1_3to3(x, seq[0].weight, seq[0].bias)
    for idx in enc.CONV3:
        wp = pack(seq[idx].weight)
        cout = int(seq[idx].weight.shape[0])
        if idx in enc.POOL_AFTER:
            y = ops.conv3x3_pool(y, wp, seq[idx].bias.detach(), cout, "replicate", True)
        else:
            y = ops.conv3x3(y, wp, seq[idx].bias.detach(), cout, False, "replicate", True)
    assert feat.dtype == torch.float32 and torch.equal(feat, y)
    seq = dec.decoder
    up = False
    for idx in dec.CONV3:
        wp = pack(seq[idx].weight)
        y = ops.conv3x3(y, wp, seq[idx].bias.detach(), int(seq[idx].weight.shape[0]), up, "replicate", idx != 28)
        up = idx in dec.UP_AFTER
    out_hw = tuple(8 * ((v + 7) // 8) for v in shape[2:])
    assert img.dtype == torch.float32 and tuple(img.shape) == (shape[0], 3) + out_hw and torch.equal(img, y)


# ---- bf16x3: hi/lo pair layout -------------------------------------------------------------------------------
@pytest.mark.parametrize("n,cin,cout,h,w,up,relu,pool", CASES)
def test_g8_pair_is_bit_identical_to_fp32_activation_bf16x3_kernels(ops, n, cin, cout, h, w, up, relu, pool):
    """bf16x3: the pair layout holds exactly the (hi, lo) split the fp32-activation kernel makes of its input, and a
    pair output is the split of the fp32 result: both directions compare bit for bit."""
    g = torch.Generator().manual_seed(cin * 3 + cout + w)
    x = torch.randn(n, cin, h, w, generator=g).cuda()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    wp = ops.pack_conv3x3_weights(wt, False, "bf16x3")
    ref = ops.conv3x3_pool(x, wp, b, cout, "replicate", relu) if pool else ops.conv3x3(x, wp, b, cout, up, "replicate", relu)
    xg = ops.to_g8_pair(x)
    assert torch.equal(ops.from_g8_pair(xg), x) or float((ops.from_g8_pair(xg) - x).abs().max()) < 1e-4
    got32 = ops.conv3x3_g8(xg, wp, b, cout, up, "replicate", relu, pool, y_g8=False)
    assert torch.equal(got32, ref)
    got = ops.conv3x3_g8(xg, wp, b, cout, up, "replicate", relu, pool, y_g8=True)
    assert ops.is_g8_pair(got) and torch.equal(got, ops.to_g8_pair(ref))
    got_b = ops.conv3x3_g8(x, wp, b, cout, up, "replicate", relu, pool, y_g8=True)     # fp32 in -> pair out
    assert torch.equal(got_b, got)


@pytest.mark.parametrize("compute", ["bf16", "bf16x3"])
@pytest.mark.parametrize("hw", [(64, 96), (70, 90)])       # (70, 90): odd sizes after the first pool, unfused pools
@pytest.mark.parametrize("sets", [[[21]], [[2, 7, 14, 19, 25]], [[21], [2, 7, 14, 19, 25]], [[4, 9, 16]]])
def test_loss_vgg_no_grad_g8_path_matches_differentiable_path(compute, hw, sets):
    """VGG19FeatureExtractor.forward_multi without gradients (content / style originals) chains uncaptured convs
    through G8 with zero padding and fused pools; with gradients it runs the fp32-activation kernels.  Every
    captured tensor must be the same bits."""
    from oracle import ref_ops as R
    from style_transfer_amd import precision
    from style_transfer_amd.loss_network import VGG19FeatureExtractor
    last = max(max(s) for s in sets)
    vgg = VGG19FeatureExtractor([last], weights=R.make_loss_vgg_state(2, upto=last)).cuda()
    g = torch.Generator().manual_seed(sum(hw))
    x = (torch.rand(2, 3, *hw, generator=g) * 4 - 2).cuda()
    with precision.compute_dtype(compute):
        with torch.no_grad():
            fast = vgg.forward_multi(x, sets)
        slow = vgg.forward_multi(x.clone().requires_grad_(True), sets)
    for fs, ss in zip(fast, slow):
        assert len(fs) == len(ss)
        for a, b in zip(fs, ss):
            assert a.dtype == torch.float32 and a.shape == b.shape
            if compute == "bf16":
                assert torch.equal(a, b.detach())
            else:
                # bf16x3: without gradients the 3-channel first conv runs the exact fp32 vector-ALU kernel, with
                # gradients the split MFMA kernel (generator._PackedWeightCache.get): fp32-grade agreement
                assert float((a - b.detach()).abs().max()) <= 1e-4 * float(b.abs().max())
