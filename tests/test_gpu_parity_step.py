"""Whole-step parity at north_star's bar (1e-3 relative in fp32), at the BASELINE configs[2] size, and a MEASURED
answer to "how far may whole-step gradients be from the oracle's": a float64 run of the oracle is the arbiter
between the two fp32 implementations (the CPU oracle in fp32 and the HIP path)."""
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu

CONTENT_IDX, STYLE_IDX = [21], [2, 7, 14, 19, 25]        # CLI defaults (train.py:80-83)
NAMES = ["content", "style", "variation", "hist", "total"]
# Whole-step gradients WITHOUT pinned switches (test_train_step_gradients_against_float64_arbiter) are reported, not
# gated on a tuned floor: test_train_step_gradients_with_pinned_gates counts the switches that differ (3 of 2.2 million
# at 48 x 48) and shows that on the same branch every gradient is within 2e-5 of float64, so the distances printed by
# the arbiter test ARE those few flips (5e-4 .. 3e-3 per flipped gate, depending on the term) plus the histogram
# term's input sensitivity (see there).  What it asserts is a sanity bound that a wrong kernel would break by orders
# of magnitude, and the loss VALUES at north_star's 1e-3.
UNPINNED_SANITY = 5e-2
RATIO = 3.0


def _rel_l2(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("use_vincent", [False, True], ids=["meanstd", "gram"])
def test_train_step_losses_at_configs2_size(use_vincent):
    """BASELINE configs[2]: batch 8 at 256x256, AdaIN + Gram + histogram + variation losses -- all five scalars of
    Trainer.compute_loss (trainer.py:246-279) within 1e-3 of the CPU oracle, for both style losses."""
    from test_gpu_train import _build
    step, enc_sd, dec_sd, vgg_sd = _build(use_vincent, CONTENT_IDX, STYLE_IDX, 0.1)
    g = torch.Generator().manual_seed(8)
    content = torch.rand(8, 3, 256, 256, generator=g)
    style = torch.rand(8, 3, 256, 256, generator=g)
    with torch.no_grad():
        ref = R.train_step_losses(content, style, enc_sd, dec_sd, vgg_sd, torch.tensor([1e-5]), torch.tensor([1e-5]),
                                  0.1, CONTENT_IDX, STYLE_IDX, 1.0, 10.0, 1.0, 10.0, use_vincent)
        got = step.forward_losses(content.cuda(), style.cuda())
    for name, a, b in zip(NAMES, got, ref):
        assert float(a) == pytest.approx(float(b), rel=1e-3), name


@pytest.mark.parametrize("use_vincent", [False, True], ids=["meanstd", "gram"])
def test_train_step_gradients_against_float64_arbiter(use_vincent):
    """Whole-step gradients cannot be compared element-wise across two fp32 implementations: a 1e-7 forward
    difference flips an isolated ReLU / max-pool / clamp gate now and then, and everything below that gate moves.
    So the distance is MEASURED: the same step in float64 (the oracle with double tensors) is the arbiter, and the
    distance of the HIP path and of the fp32 CPU oracle to it is printed per checked parameter and loss term (run with
    -s).  These unpinned distances are explained, not tuned away: test_train_step_gradients_with_pinned_gates counts
    the flipped switches and bounds the same-branch distance at 1e-4; here only a sanity bound is asserted.  Loss values: within
    1e-3 (north_star) of the float64 values, or RATIO x the fp32 oracle's own distance where fp32 itself cannot do
    better (the Gram loss at this size: 36 pixels for 512 channels, a rank-deficient covariance whose eigenvalue
    clips put the fp32 CPU oracle 1.6e-3 from float64; at the configs[2] size both are within 1e-3, test above)."""
    from test_gpu_train import _build
    eps_limit = 0.1
    step, enc_sd, dec_sd, vgg_sd = _build(use_vincent, CONTENT_IDX, STYLE_IDX, eps_limit)
    g = torch.Generator().manual_seed(5)
    content = torch.rand(2, 3, 48, 48, generator=g)
    style = torch.rand(2, 3, 48, 48, generator=g)
    keys = ("decoder.28.weight", "decoder.28.bias", "decoder.21.weight", "decoder.14.weight", "decoder.5.weight",
            "decoder.1.weight", "decoder.1.bias")

    def oracle(dtype, i):
        to = lambda sd: {k: v.to(dtype) for k, v in sd.items()}       # noqa: E731
        dec = {k: v.to(dtype).clone().requires_grad_(True) for k, v in dec_sd.items()}
        ce = torch.tensor([1e-5], dtype=dtype, requires_grad=True)
        se = torch.tensor([1e-5], dtype=dtype, requires_grad=True)
        with R.hist_kernel_dtype(dtype):
            out = R.train_step_losses(content.to(dtype), style.to(dtype), to(enc_sd), dec, to(vgg_sd), ce, se,
                                      eps_limit, CONTENT_IDX, STYLE_IDX, 1.0, 10.0, 1.0, 10.0, use_vincent)
        out[i].backward()
        return [float(v.detach()) for v in out], {k: dec[k].grad for k in keys}, (float(ce.grad), float(se.grad))

    report, eps_report, loss_report = [], [], []
    for i, name in enumerate(NAMES):
        l64, g64, e64 = oracle(torch.float64, i)
        l32, g32, e32 = oracle(torch.float32, i)
        for p in list(step.decoder.parameters()) + list(step.transformer.parameters()):
            p.grad = None
        losses = step.forward_losses(content.cuda(), style.cuda())
        losses[i].backward()
        loss_report.append((name, float(losses[i].detach()), l32[i], l64[i]))
        for k in keys:
            if name == "variation" and k.endswith("28.bias"):
                continue                                     # analytically zero
            mod = step.decoder.decoder[int(k.split(".")[1])]
            got = mod.weight.grad if k.endswith("weight") else mod.bias.grad
            report.append((name, k, _rel_l2(got, g64[k]), _rel_l2(g32[k], g64[k])))
        scale = max(abs(e64[0]), abs(e64[1]), 1e-6)
        for which, got, a64, a32 in (("content_eps", float(step.transformer.content_eps.grad), e64[0], e32[0]),
                                     ("style_eps", float(step.transformer.style_eps.grad), e64[1], e32[1])):
            eps_report.append((name, which, abs(got - a64) / scale, abs(a32 - a64) / scale))
    tag = "gram" if use_vincent else "meanstd"
    print(f"\nfloat64 arbiter, {tag}: relative distance to the float64 step (HIP | fp32 CPU oracle)")
    for name, got, l32, l64 in loss_report:
        print(f"  loss {name:9s} {abs(got - l64) / abs(l64):.2e} | {abs(l32 - l64) / abs(l64):.2e}")
    for name, k, e_hip, e_o32 in report + eps_report:
        print(f"  grad {name:9s} {k:18s} {e_hip:.2e} | {e_o32:.2e}")
    for name, got, l32, l64 in loss_report:
        assert abs(got - l64) <= max(RATIO * abs(l32 - l64), 1e-3 * abs(l64)), ("loss", name, got, l32, l64)
    for name, k, e_hip, e_o32 in report + eps_report:
        # The histogram term (and the total that contains it) is REPORTED only: unpinned it moves by percents with
        # any last-bit change of the decoded image (1 / (x + 1e-6) on pixels just above the clamp; measured 2e-2 with
        # the direct upsample form, 7e-2 with the sub-pixel form, the fp32 CPU oracle itself 1-2e-2) -- its bar is
        # in the pinned test (block at identical input: 2e-5).
        if name not in ("hist", "total"):
            assert e_hip <= UNPINNED_SANITY, (name, k, e_hip, e_o32)


def _hip_gates(step, acts, d_n):
    """The switches the HIP step took, under the oracle's names (oracle.ref_ops.gates): decoder ReLU masks from the
    chain's own activations, loss-VGG ReLU masks and max-pool winners from the same kernels run layer by layer on the
    step's normalised decoded image, the [0, 1] clamp mask of the histogram block."""
    import torch.nn.functional as F
    g = {}
    for idx, a in zip(step.decoder.CONV3[:-1], acts[:-1]):
        g[f"dec.{idx}"] = (a > 0).cpu()
    last = max(STYLE_IDX)
    with torch.no_grad():
        outs = step.loss_vgg.forward_multi(d_n, [list(range(last + 1))])[0]
    layout = step.loss_vgg.layout
    for idx in range(last + 1):
        kind = layout[idx][0]
        if kind == "relu":
            g[f"vgg.relu.{idx}"] = (outs[idx] > 0).cpu()            # the in-place ReLU's output (== the conv capture)
        elif kind == "pool":
            _, win = F.max_pool2d(outs[idx - 1].cpu(), 2, 2, return_indices=True)
            g[f"vgg.pool.{idx}"] = win
    dn = d_n.detach().cpu()
    g["hist"] = ((dn >= 0) & (dn <= 1), dn > 1)
    return g


def _count_flips(a, b):
    """Positions where two recorded switch sets differ, per name."""
    out = {}
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, tuple):
            n = int((x[0] != y[0]).sum() + (x[1] != y[1]).sum())
        else:
            n = int((x != y).sum())
        if n:
            out[k] = n
    return out


@pytest.mark.parametrize("style", ["meanstd", "gram-newton-schulz", "gram-eigh"])
def test_train_step_gradients_with_pinned_gates(style, monkeypatch):
    """measure the claim instead of absorbing it.  (i) The ReLU / max-pool / clamp switches of the
    HIP step and of the fp32 CPU oracle are exported and compared: they differ in a handful of isolated positions
    (printed).  (ii) With the HIP step's switches FORCED onto the oracle (oracle.ref_ops.gates) both implementations
    differentiate the same piecewise-linear branch, and every checked gradient of every loss term is within 1e-4
    relative L2 of the float64 oracle -- no floor, no ratio.  (mean/std style loss; the Gram loss adds eigenvalue
    clips, measured in the arbiter test above.)"""
    from test_gpu_train import _build
    from style_transfer_amd.autograd_conv import DecoderChainFunction
    from style_transfer_amd.losses import VincentStyleLoss
    eps_limit = 0.1
    use_vincent = style != "meanstd"
    if use_vincent:
        monkeypatch.setattr(VincentStyleLoss, "STYLE_ROOT", style[len("gram-"):])
    step, enc_sd, dec_sd, vgg_sd = _build(use_vincent, CONTENT_IDX, STYLE_IDX, eps_limit)
    g = torch.Generator().manual_seed(5)
    # Gram: 96 x 96, so that the deepest style layers have 144 pixels for 512 channels -- still rank-deficient (the
    # reference's eigenvalue clips are switches of their own, NOT pinned here), but not the 36-pixel extreme
    side = 96 if use_vincent else 48
    content = torch.rand(2, 3, side, side, generator=g)
    style = torch.rand(2, 3, side, side, generator=g)
    keys = ("decoder.28.weight", "decoder.28.bias", "decoder.21.weight", "decoder.14.weight", "decoder.5.weight",
            "decoder.1.weight", "decoder.1.bias")

    def oracle(dtype, i, mode, store, decoded=None):
        to = lambda sd: {k: v.to(dtype) for k, v in sd.items()}       # noqa: E731
        dec = {k: v.to(dtype).clone().requires_grad_(True) for k, v in dec_sd.items()}
        ce = torch.tensor([1e-5], dtype=dtype, requires_grad=True)
        se = torch.tensor([1e-5], dtype=dtype, requires_grad=True)
        with R.hist_kernel_dtype(dtype), R.gates(mode, store):
            out = R.train_step_losses(content.to(dtype), style.to(dtype), to(enc_sd), dec, to(vgg_sd), ce, se,
                                      eps_limit, CONTENT_IDX, STYLE_IDX, 1.0, 10.0, 1.0, 10.0, use_vincent,
                                      decoded_override=decoded)
        out[i].backward()
        return {k: dec[k].grad for k in keys}, (float(ce.grad), float(se.grad))

    # the HIP step's switches (one forward; they do not depend on which loss term is differentiated)
    DecoderChainFunction.TAP = acts = []
    try:
        losses = step.forward_losses(content.cuda(), style.cuda())
    finally:
        DecoderChainFunction.TAP = None
    assert len(acts) == len(step.decoder.CONV3)
    with torch.no_grad():
        d_n = step.norm(acts[-1])
    hip = _hip_gates(step, acts, d_n)
    own32, own64 = {}, {}
    oracle(torch.float32, 4, "record", own32)
    oracle(torch.float64, 4, "record", own64)
    assert set(own32) == set(hip), (sorted(set(own32) ^ set(hip)))
    total = sum(int(v[0].numel()) if isinstance(v, tuple) else int(v.numel()) for v in hip.values())
    f_hip32, f_hip64, f_3264 = _count_flips(hip, own32), _count_flips(hip, own64), _count_flips(own32, own64)
    print(f"\nswitches of the step: {total} in {len(hip)} tensors; positions that differ --")
    print(f"  HIP vs fp32 oracle    : {sum(f_hip32.values())}  {f_hip32}")
    print(f"  HIP vs float64 oracle : {sum(f_hip64.values())}  {f_hip64}")
    print(f"  fp32 vs float64 oracle: {sum(f_3264.values())}  {f_3264}")
    # (i) isolated flips only: well under one switch in 10^4 (measured: a few units out of ~3.6 million)
    assert sum(f_hip32.values()) <= max(8, total // 100000), f_hip32
    assert sum(f_hip64.values()) <= max(8, total // 100000), f_hip64

    # (ii) same branch on both sides: plain rounding is all that is left
    rows = []
    for i, name in enumerate(NAMES):
        # hist / total: everything after the decoder is evaluated at the HIP step's own decoded image (see
        # train_step_losses), so that the comparison is at identical input
        at = acts[-1].cpu() if name in ("hist", "total") else None
        g64, e64 = oracle(torch.float64, i, "force", hip, at)
        g32, e32 = oracle(torch.float32, i, "force", hip, at)
        for p in list(step.decoder.parameters()) + list(step.transformer.parameters()):
            p.grad = None
        losses = step.forward_losses(content.cuda(), style.cuda())
        losses[i].backward()
        for k in keys:
            if name == "variation" and k.endswith("28.bias"):
                continue                                     # analytically zero
            mod = step.decoder.decoder[int(k.split(".")[1])]
            got = mod.weight.grad if k.endswith("weight") else mod.bias.grad
            rows.append((name, k, _rel_l2(got, g64[k]), _rel_l2(g32[k], g64[k])))
        scale = max(abs(e64[0]), abs(e64[1]), 1e-30)
        rows.append((name, "content_eps", abs(float(step.transformer.content_eps.grad) - e64[0]) / scale, abs(e32[0] - e64[0]) / scale))
        rows.append((name, "style_eps", abs(float(step.transformer.style_eps.grad) - e64[1]) / scale, abs(e32[1] - e64[1]) / scale))
    print("pinned switches: relative distance to the float64 oracle on the same branch (HIP | fp32 CPU oracle)")
    for name, k, e_hip, e_o32 in rows:
        print(f"  grad {name:9s} {k:18s} {e_hip:.2e} | {e_o32:.2e}")
    for name, k, e_hip, e_o32 in rows:
        if name in ("content", "variation") or (name == "style" and not use_vincent):
            assert e_hip <= 1e-4, (name, k, e_hip, e_o32)      # measured 2e-7 .. 2e-5
        elif name == "style":
            # Gram: clip(0) / clip(0.1) on the eigenvalues of rank-deficient matrices are switches too, and fp32
            # eigenvalues of the null space are rounding noise in ANY implementation (the reference's fp32 included)
            assert e_hip <= 5.0 * e_o32 + 1e-3, (name, k, e_hip, e_o32)
        else:
            # The histogram term (and the total that contains it): d log(x + 1e-6) / dx = 1 / (x + 1e-6) turns the
            # 1e-7 forward rounding of a decoded pixel that sits just above the clamp into a percent-level change of
            # its gradient (a random-weight decoder leaves a few such pixels: evaluated each at its OWN decoded image
            # the two fp32 implementations are 2e-2 .. 7e-2 apart from float64, the reference's fp32 arithmetic
            # included).  The oracle is therefore evaluated at the HIP step's decoded image for these two terms --
            # same input, same branch.  Measured: weight / bias gradients 1e-5 .. 1e-4 (the fp32 CPU oracle: 1e-4 ..
            # 2.4e-4, i.e. the HIP path is the closer one), the two eps scalars up to 2.2e-4 (sums over the whole
            # feature map with cancellation between the terms; fp32 oracle up to 1.7e-3).
            assert e_hip <= (1e-3 if k.endswith("_eps") else 2e-4), (name, k, e_hip, e_o32)

    # the histogram term at IDENTICAL input: HistLoss (0.4 style + 0.6 content, trainer.py:269-271) on the HIP step's
    # own normalised decoded image against the float64 oracle on the same tensor and the same clamp mask
    with torch.no_grad():
        c_n, s_n = step.norm(content.cuda()), step.norm(style.cuda())
    x = d_n.detach().clone().requires_grad_(True)
    (0.4 * step.hist_loss(x, s_n) + 0.6 * step.hist_loss(x, c_n)).backward()
    x64 = d_n.detach().cpu().double().requires_grad_(True)
    with R.hist_kernel_dtype(torch.float64), R.gates("force", {"hist": hip["hist"]}):
        (0.4 * R.hist_loss(x64, s_n.cpu().double(), gate_name="hist")
         + 0.6 * R.hist_loss(x64, c_n.cpu().double(), gate_name="hist")).backward()
    e_blk = _rel_l2(x.grad, x64.grad)
    print(f"  histogram block at identical input: d loss / d image {e_blk:.2e} from float64")
    assert e_blk <= 1e-4, e_blk
