"""GPU parity of the loss VGG's differentiated pass in bf16 mode as one autograd node whose forward and backward hand
activations / gradients over in G8 (autograd_vgg.VGGChainFunction, st_conv3x3_dgrad_bf16_g8, st_grad_join_g8).

The bar is BIT-IDENTITY with the per-layer autograd graph of the fp32-activation bf16 kernels (which
tests/test_gpu_bf16.py pins to the same-rounding oracle): a G8 value is the round-to-nearest-even the consumer's
staging applies to an fp32 value anyway, the chunk/tap accumulation order of the data-gradient kernel is the same,
a ReLU gate commutes with the rounding, and the join kernel adds the same fp32 terms the autograd engine adds.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from style_transfer_amd import ops as o
    return o


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("n,c,h,w", [(2, 64, 32, 48), (1, 8, 7, 9), (3, 128, 5, 12)])
@pytest.mark.parametrize("terms", ["chain", "chain+post", "chain+post+pre", "post", "pre", "chain+pre", "chain-nogate"])
def test_grad_join_plain(ops, n, c, h, w, terms):
    shape = (n, c, h, w)
    chain = _rand(shape, 1) if "chain" in terms else None
    post = _rand(shape, 2) if "post" in terms else None
    pre = _rand(shape, 3) if "pre" in terms else None
    x = None if terms in ("pre", "chain-nogate") else _rand(shape, 4)
    got = ops.grad_join_g8(shape, chain=chain, x_gate=x, add_post=post, add_pre=pre)
    s = torch.zeros(shape, device="cuda")
    if chain is not None:
        s = chain.clone()
    if post is not None:
        s = s + post if chain is not None else post.clone()
    if x is not None:
        s = torch.where(x > 0, s, torch.zeros_like(s))
    if pre is not None:
        s = s + pre if (chain is not None or post is not None) else pre.clone()
    assert ops.is_g8(got)
    assert torch.equal(got, ops.to_g8(s))


@pytest.mark.parametrize("n,c,h,w", [(2, 64, 32, 48), (1, 16, 6, 10), (1, 128, 64, 64)])
@pytest.mark.parametrize("terms", ["pooled", "pooled+post", "pooled+post+pre"])
def test_grad_join_through_the_pool(ops, n, c, h, w, terms):
    shape = (n, c, h, w)
    x = torch.relu(_rand(shape, 5))                       # many exact ties at 0, like a ReLU output
    x[:, :, :2, :2] = 0.0                                 # an all-zero window
    pooled = _rand((n, c, h // 2, w // 2), 6)
    post = _rand(shape, 7) if "post" in terms else None
    pre = _rand(shape, 8) if "pre" in terms else None
    got = ops.grad_join_g8(shape, chain_pooled=pooled, x_gate=x, add_post=post, add_pre=pre)
    s = ops.maxpool2x2_bwd(x, pooled, False)              # the stand-alone kernel (tests/test_gpu_backward.py)
    if post is not None:
        s = s + post
    s = torch.where(x > 0, s, torch.zeros_like(s))
    if pre is not None:
        s = s + pre
    assert torch.equal(got, ops.to_g8(s))
    # ... and torch's own routing on tie-free data
    xr = _rand(shape, 9).requires_grad_(True)
    torch.nn.functional.max_pool2d(xr, 2).backward(pooled)
    got = ops.grad_join_g8(shape, chain_pooled=pooled, x_gate=xr.detach())
    assert torch.equal(got, ops.to_g8(torch.where(xr.detach() > 0, xr.grad, torch.zeros_like(xr.grad))))


DGRAD_CASES = [
    # n, cout (channels of dy), cin (channels of dx), h, w
    (2, 64, 64, 40, 40),
    (1, 128, 64, 33, 47),
    (1, 256, 128, 24, 40),
    (1, 512, 512, 8, 8),
    (3, 64, 64, 320, 320),      # persistent G8 -> G8 form (> 256 tiles)
    (2, 64, 3, 48, 64),         # the first layer: a 3-channel fp32 gradient
]


@pytest.mark.parametrize("n,cout,cin,h,w", DGRAD_CASES)
def test_dgrad_g8_is_bit_identical_to_the_gated_fp32_form(ops, n, cout, cin, h, w):
    g = torch.Generator().manual_seed(cout + cin + h)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).cuda()
    wpt = ops.pack_conv3x3_weights(wt, True, "bf16")
    dy = _rand((n, cout, h, w), 11)
    y = torch.relu(_rand((n, cout, h, w), 12))
    ref = ops.conv3x3_dgrad(dy, y, wpt, cin, False, ops.PAD_ZERO)            # gate applied while staging
    dyg = ops.grad_join_g8(dy.shape, chain=dy, x_gate=y)
    got = ops.conv3x3_dgrad_g8(dyg, wpt, cin, dx_g8=False)
    assert torch.equal(got, ref)
    # fp32 dy without a gate
    assert torch.equal(ops.conv3x3_dgrad_g8(dy, wpt, cin, dx_g8=False), ops.conv3x3_dgrad(dy, None, wpt, cin, False,
                                                                                        ops.PAD_ZERO))
    if cin % 8 == 0:
        below = torch.relu(_rand((n, cin, h, w), 13))
        got8 = ops.conv3x3_dgrad_g8(dyg, wpt, cin, dx_g8=True)
        assert torch.equal(got8, ops.to_g8(ref))
        gated = ops.conv3x3_dgrad_g8(dyg, wpt, cin, dx_g8=True, dx_gate=ops.to_g8(below))
        # the gate is the sign of the bf16 value the layer below handed over
        keep = ops.from_g8(ops.to_g8(below)) > 0
        assert torch.equal(gated, ops.to_g8(torch.where(keep, ref, torch.zeros_like(ref))))


def _extractor(sets, seed=3):
    import warnings
    from style_transfer_amd.loss_network import VGG19FeatureExtractor
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = VGG19FeatureExtractor(sorted({i for s in sets for i in s}), weights=None, seed=seed).cuda()
    return net


@pytest.mark.parametrize("sets,n,h,w", [
    ([[21], [2, 7, 14, 19, 25]], 2, 64, 64),          # the CLI defaults (trainer.py:446-457)
    ([[21], [2, 7, 14, 19, 25]], 1, 72, 88),          # 9 x 11 at the third pool: the odd-size route
    ([[4, 8], [1, 6, 11]], 2, 32, 48),                # captured pool output, captured ReLU indices, pre-ReLU 11
    ([[25], [21, 25]], 1, 32, 32),                    # one conv captured both before and after its ReLU
    ([[7]], 1, 40, 24),                               # a single tap; nothing captured below it
])
def test_chain_matches_the_per_layer_graph(sets, n, h, w):
    from style_transfer_amd import precision
    net = _extractor(sets)
    x0 = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(5)).cuda()
    weights = [torch.randn(1, generator=torch.Generator().manual_seed(40 + k)).item() for k in range(16)]

    def run(chain):
        net.GRAD_CHAIN = chain
        x = x0.clone().requires_grad_(True)
        with precision.compute_dtype("bf16"):
            outs = net.forward_multi(x, sets)
            flat = [t for s in outs for t in s]
            loss = sum(wt * (t * t).mean() + 0.1 * wt * t.mean() for wt, t in zip(weights, flat))
            loss.backward()
        return [t.detach() for t in flat], x.grad

    try:
        ref_out, ref_grad = run(False)
        got_out, got_grad = run(True)
    finally:
        net.GRAD_CHAIN = True
    for a, b in zip(got_out, ref_out):
        assert torch.equal(a, b)
    assert float(ref_grad.abs().max()) > 0
    assert torch.equal(got_grad, ref_grad)


def test_chain_with_a_tap_that_carries_no_gradient():
    from style_transfer_amd import precision
    sets = [[21], [2, 7, 14]]
    net = _extractor(sets)
    x0 = torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(6)).cuda()

    def run(chain):
        net.GRAD_CHAIN = chain
        x = x0.clone().requires_grad_(True)
        with precision.compute_dtype("bf16"):
            outs = net.forward_multi(x, sets)
            (outs[1][1] ** 2).mean().backward()           # only relu2_2: the layers above it see no gradient
        return x.grad
    try:
        ref, got = run(False), run(True)
    finally:
        net.GRAD_CHAIN = True
    assert torch.equal(got, ref)


@pytest.mark.parametrize("n,c,h,w", [(2, 64, 32, 48), (1, 16, 2, 2), (3, 128, 70, 36)])
def test_maxpool_to_g8(ops, n, c, h, w):
    x = _rand((n, c, h, w), 21)
    assert torch.equal(ops.maxpool2x2_g8(x), ops.to_g8(ops.maxpool2x2(x, False)))


def test_no_grad_g8_pass_with_the_pool_written_in_g8():
    """loss_network._forward_multi_g8 (style / content originals): captured layer -> pool -> conv, the pool output
    goes to the next conv in G8; same bits as the per-layer fp32-activation pass."""
    from style_transfer_amd import precision
    sets = [[21], [2, 7, 14, 19, 25]]
    net = _extractor(sets)
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(8)).cuda()
    with precision.compute_dtype("bf16"), torch.no_grad():
        got = net.forward_multi(x, sets)
    net.GRAD_CHAIN = False
    try:
        with precision.compute_dtype("bf16"):
            ref = net.forward_multi(x.clone().requires_grad_(True), sets)      # per-layer graph, fp32 activations
    finally:
        net.GRAD_CHAIN = True
    for a, b in zip([t for s in got for t in s], [t for s in ref for t in s]):
        assert torch.equal(a, b.detach())


@pytest.mark.parametrize("n,c,h,w", [(2, 64, 32, 48), (1, 8, 7, 9), (1, 128, 64, 64)])
@pytest.mark.parametrize("terms", ["post", "pre", "chain+post", "pooled+post+pre", "chain+post-tensor+pre"])
def test_grad_join_with_affine_loss_coefficients(ops, n, c, h, w, terms):
    """st_grad_join_g8_affine: the mean/std loss gradient as (k0, k1, mean) per plane, evaluated inside the join, is
    the tensor st_meanstd_loss_bwd writes -- same fp32 expression, so the G8 results are the same bits."""
    from style_transfer_amd import losses
    if "pooled" in terms:
        h, w = h + h % 2, w + w % 2                                   # the pooled form takes even sizes only
    shape = (n, c, h, w)
    x = torch.relu(_rand(shape, 11)) if "pooled" in terms else _rand(shape, 11)
    target = _rand(shape, 12, 1.5) + 0.25
    tap = x.clone().requires_grad_(True)
    (losses.AdaINStyleLoss()(tap, target) * 3.5).backward()          # the tensor form, g = 3.5
    dense = tap.grad
    ma, sa = ops.mean_std(x, 1e-5)
    mb, sb = ops.mean_std(target, 1e-5)
    coeffs = ops.meanstd_loss_coeffs(ma, sa, mb, sb, torch.tensor(3.5, device="cuda"), h * w)
    assert tuple(coeffs.shape) == (n * c, 4) and torch.equal(coeffs[:, 2], ma.reshape(-1))
    chain = _rand(shape, 13) if "chain" in terms else None
    pooled = _rand((n, c, h // 2, w // 2), 14) if "pooled" in terms else None
    other = _rand(shape, 15)
    kw = dict(chain=chain, chain_pooled=pooled, x_gate=x)
    if terms == "post":
        want = ops.grad_join_g8(shape, add_post=dense, **kw)
        got = ops.grad_join_g8(shape, aff_x=x, aff_post=coeffs, **kw)
    elif terms == "pre":
        want = ops.grad_join_g8(shape, add_pre=dense, **kw)
        got = ops.grad_join_g8(shape, aff_x=x, aff_pre=coeffs, **kw)
    elif terms == "chain+post":
        want = ops.grad_join_g8(shape, add_post=dense, **kw)
        got = ops.grad_join_g8(shape, aff_x=x, aff_post=coeffs, **kw)
    elif terms == "pooled+post+pre":
        want = ops.grad_join_g8(shape, add_post=dense, add_pre=other, **kw)
        got = ops.grad_join_g8(shape, aff_x=x, aff_post=coeffs, add_pre=other, **kw)
    else:                                                             # a tensor term beside an affine one
        want = ops.grad_join_g8(shape, add_post=other, add_pre=dense, **kw)
        got = ops.grad_join_g8(shape, add_post=other, aff_x=x, aff_pre=coeffs, **kw)
    assert float(ops.from_g8(want).abs().max()) > 0
    assert torch.equal(got, want)


@pytest.mark.parametrize("n,h,w", [(2, 64, 64), (1, 72, 88)])
def test_affine_tap_handover_is_bit_identical_to_the_tensor_form(n, h, w):
    """The CLI's loss taps with AdaINStyleLoss on them (trainer.py:446-457): coefficients handed to the join
    (loss_network.AFFINE_TAPS) vs. the [n, c, h, w] loss-gradient tensors."""
    from style_transfer_amd import losses, precision
    sets = [[21], [2, 7, 14, 19, 25]]
    net = _extractor(sets)
    x0 = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(8)).cuda()
    style = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(9)).cuda()
    crit = losses.AdaINStyleLoss()

    def run(affine):
        net.AFFINE_TAPS = affine
        x = x0.clone().requires_grad_(True)
        with precision.compute_dtype("bf16"):
            with torch.no_grad():
                targets = net.forward_multi(style, sets)
            outs = net.forward_multi(x, sets)
            assert all(getattr(t, "_st_chain_tap", False) == affine for t in outs[1])
            loss = (outs[0][0] - targets[0][0]).pow(2).mean()
            for k, (a, b) in enumerate(zip(outs[1], targets[1])):
                loss = loss + (0.5 + k) * crit(a, b)
            loss.backward()
        return loss.detach(), x.grad

    try:
        ref_loss, ref = run(False)
        got_loss, got = run(True)
    finally:
        del net.AFFINE_TAPS                                            # back to the class default
    assert float(ref.abs().max()) > 0
    assert torch.equal(got_loss, ref_loss)
    assert torch.equal(got, ref)


# ---- pool-input layers nobody captures: conv + ReLU + pool in one launch, winner codes for the backward ---------------
def _codes_ref(y_full):
    """(position of the first maximum in scan order, maximum > 0) per 2 x 2 window, in G8 order [n, c/8, h/2, w/2, 8]"""
    n, c, h, w = y_full.shape
    win = y_full.reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
    top = win.max(-1, keepdim=True).values
    eq = win == top
    first = eq & (eq.cumsum(-1) == 1)
    arg = first.to(torch.uint8).argmax(-1)
    pos = top[..., 0] > 0
    g8 = lambda t: t.reshape(n, c // 8, 8, h // 2, w // 2).permute(0, 1, 3, 4, 2).contiguous()   # noqa: E731
    return g8(arg), g8(pos)


@pytest.mark.parametrize("n,cin,cout,h,w,x_g8", [
    (2, 64, 64, 32, 64, False),        # conv1_2's form: fp32 tap in, 64-channel tile
    (1, 64, 128, 48, 40, True),        # partial pixel tiles
    (2, 256, 256, 16, 32, True),       # conv3_4's form
    (1, 16, 32, 34, 6, True),          # the 32-channel tile, rows past the tile
    (1, 128, 48, 20, 36, False),       # a channel tail inside the last 64-channel tile
])
def test_conv_pool_codes_forward_and_backward(ops, n, cin, cout, h, w, x_g8):
    x = torch.relu(_rand((n, cin, h, w), 51))
    wt = _rand((cout, cin, 3, 3), 52, 0.05)
    bias = _rand((cout,), 53, 0.1)
    wp = ops.pack_conv3x3_weights(wt, False, "bf16")
    xin = ops.to_g8(x) if x_g8 else x
    y_full = ops.conv3x3_g8(xin, wp, bias, cout, False, ops.PAD_ZERO, relu=True, pool=False, y_g8=False)
    y_pool, codes = ops.conv3x3_pool_g8_codes(xin, wp, bias, cout, ops.PAD_ZERO, relu=True)
    assert torch.equal(y_pool, ops.maxpool2x2_g8(y_full))
    arg, pos = _codes_ref(y_full)
    assert torch.equal((codes & 4) != 0, pos)
    assert int((codes > 7).sum()) == 0
    # the position only matters (and is only defined by the ReLU output) where the maximum is positive
    assert torch.equal((codes & 3)[pos], arg[pos])
    g = ops.to_g8(_rand((n, cout, h // 2, w // 2), 54))
    got = ops.grad_pool_codes_g8(g, codes)
    ref = ops.grad_join_g8((n, cout, h, w), chain_pooled=ops.from_g8(g), x_gate=y_full)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("sets,n,h,w", [
    ([[21], [2, 7, 14, 19, 25]], 2, 64, 64),          # the CLI defaults: conv3_4 is the one uncaptured pool input
    ([[21], [14, 19, 25]], 1, 64, 96),                # conv1_2 and conv2_2 uncaptured as well
    ([[21], [1, 6, 25]], 1, 72, 88),                  # 9 x 11 at the third pool: that layer keeps the odd-size route
])
def test_pool_codes_chain_is_bit_identical_to_the_fp32_pool_input_form(sets, n, h, w):
    from style_transfer_amd import autograd_vgg, precision
    net = _extractor(sets)
    x0 = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(9)).cuda()
    weights = [torch.randn(1, generator=torch.Generator().manual_seed(60 + k)).item() for k in range(16)]

    def run(codes):
        autograd_vgg.POOL_CODES = codes
        x = x0.clone().requires_grad_(True)
        with precision.compute_dtype("bf16"):
            outs = net.forward_multi(x, sets)
            flat = [t for s in outs for t in s]
            loss = sum(wt * (t * t).mean() + 0.1 * wt * t.mean() for wt, t in zip(weights, flat))
            loss.backward()
        return [t.detach() for t in flat], x.grad

    try:
        ref_out, ref_grad = run(False)
        got_out, got_grad = run(True)
    finally:
        autograd_vgg.POOL_CODES = True
    for a, b in zip(got_out, ref_out):
        assert torch.equal(a, b)
    assert float(ref_grad.abs().max()) > 0
    assert torch.equal(got_grad, ref_grad)


# ---- a loss tap that feeds a pool: fp32 tap + pooled G8 from one launch ----------------------------------------------
@pytest.mark.parametrize("n,cin,cout,h,w", [
    (2, 64, 64, 32, 64),          # conv1_2's form (the 64-channel tile)
    (1, 128, 128, 48, 40),        # conv2_2's form, partial pixel tiles
    (1, 16, 32, 34, 6),           # the 32-channel tile
    (1, 64, 48, 20, 36),          # a channel tail inside the last tile
])
def test_conv_tap_pool_dual_output(ops, n, cin, cout, h, w):
    x = ops.to_g8(torch.relu(_rand((n, cin, h, w), 71)))
    wt = _rand((cout, cin, 3, 3), 72, 0.05)
    bias = _rand((cout,), 73, 0.1)
    wp = ops.pack_conv3x3_weights(wt, False, "bf16")
    ref = ops.conv3x3_g8(x, wp, bias, cout, False, ops.PAD_ZERO, relu=True, pool=False, y_g8=False)
    y, yp = ops.conv3x3_g8_tap_pool(x, wp, bias, cout, ops.PAD_ZERO, relu=True)
    assert torch.equal(y, ref)
    assert torch.equal(yp, ops.maxpool2x2_g8(ref))


def test_tap_pool_passes_are_bit_identical_to_the_separate_pool():
    from style_transfer_amd import autograd_vgg, precision
    sets = [[21], [2, 7, 14, 19, 25]]
    net = _extractor(sets)
    x0 = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(10)).cuda()

    def run(tap_pool):
        autograd_vgg.TAP_POOL = tap_pool
        with precision.compute_dtype("bf16"), torch.no_grad():
            plain = [t for s in net.forward_multi(x0, sets) for t in s]
        x = x0.clone().requires_grad_(True)
        with precision.compute_dtype("bf16"):
            flat = [t for s in net.forward_multi(x, sets) for t in s]
            sum((t * t).mean() for t in flat).backward()
        return plain, [t.detach() for t in flat], x.grad

    try:
        ref = run(False)
        got = run(True)
    finally:
        autograd_vgg.TAP_POOL = True
    for a, b in zip(got[0] + got[1], ref[0] + ref[1]):
        assert torch.equal(a, b)
    assert torch.equal(got[2], ref[2])
