"""Whole-step parity at north_star's bar (1e-3 relative in fp32), at the BASELINE configs[2] size, and a MEASURED
answer to "how far may whole-step gradients be from the oracle's": a float64 run of the oracle is the arbiter
between the two fp32 implementations (the CPU oracle in fp32 and the HIP path)."""
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu

CONTENT_IDX, STYLE_IDX = [21], [2, 7, 14, 19, 25]        # CLI defaults (train.py:80-83)
NAMES = ["content", "style", "variation", "hist", "total"]
# Measured on this input (table printed with -s): where no gate flips both fp32 implementations sit 2e-7..3e-6 from
# float64; ONE flipped ReLU gate (the HIP path flips one on this input where the CPU oracle flips none -- on other
# inputs it is the other way round) moves the weight gradients below it by 5e-4 (content / style terms) to 2.6e-3
# (variation term, whose gradient is concentrated in few pixels) in relative L2; the histogram and total terms are
# 1-2.4e-2 from float64 in BOTH fp32 implementations (clamp mask, see HIST_EPS_FLOOR).
FLIP_FLOOR = 5e-3
RATIO = 3.0
# d hist / d eps sums a gradient image whose mask is the [0,1] clamp of losses.py:170 over a decoded image that is
# ~96 % clamped with random weights: pixels that enter or leave the clamp between two fp32 implementations move it by
# percents (the fp32 CPU oracle itself is 1.6 % from float64 on this input).
HIST_EPS_FLOOR = 1e-1


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
    HIP path has to be as close to it as the fp32 CPU oracle is -- err(HIP) <= RATIO x err(oracle fp32) per checked
    parameter and loss term, with a floor of FLIP_FLOOR for the cases where the fp32 oracle happens to flip no
    gate against float64 and the HIP path flips one.  The full table is printed (run with -s).  Loss values: within
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
    for name, k, e_hip, e_o32 in report:
        assert e_hip <= max(RATIO * e_o32, FLIP_FLOOR), (name, k, e_hip, e_o32)
    for name, k, e_hip, e_o32 in eps_report:
        floor = HIST_EPS_FLOOR if name in ("hist", "total") else FLIP_FLOOR
        assert e_hip <= max(RATIO * e_o32, floor), (name, k, e_hip, e_o32)
