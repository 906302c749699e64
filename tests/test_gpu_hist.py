"""GPU parity of the RGB-uv histogram / HistLoss kernels against reference-generated golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("method", ["inverse-quadratic", "RBF", "thresholding"])
@pytest.mark.parametrize("iscale", [True, False])
def test_hist_block_golden(golden_dir, method, iscale):
    from style_transfer_amd.losses import RGBuvHistBlock
    g = np.load(os.path.join(golden_dir, "hist.npz"))
    blk = RGBuvHistBlock(insz=128, h=64, intensity_scale=iscale, method=method, hist_boundary=[-3, 3])
    tag = f"{method}_{int(iscale)}"
    for which in ("small", "big"):
        h = blk(torch.from_numpy(g[which]).cuda())
        ref = torch.from_numpy(g[f"hist_{which}_{tag}"])
        assert h.shape == ref.shape
        np.testing.assert_allclose(h.sum(dim=(1, 2, 3)).cpu(), ref.sum(dim=(1, 2, 3)), rtol=1e-5)
        if method == "thresholding":
            # Hard bins: a pixel whose log-chroma sits within an ulp of a bin edge may flip between torch-CPU log
            # and device logf; everything else is exact.  Stated precisely, in units of the
            # un-normalised histogram (x total mass T, counted directly): every bin that differs differs by ONE
            # pixel's weight -- 1 count with intensity_scale=False, at most Iy <= sqrt(3 + 1e-6) with it -- all
            # other bins agree to the rescaling a moved pixel causes (<< one count), and only a handful move.
            x = torch.from_numpy(g[which])[:, :3].float().clamp(0, 1)
            if x.shape[2] > 128 or x.shape[3] > 128:
                x = torch.nn.functional.interpolate(x, size=(128, 128), mode="bilinear", align_corners=False)
            total = _raw_counts(x, iscale).double().sum(dim=(1, 2, 3), keepdim=True)
            d = (h.cpu().double() - ref.double()).abs() * total
            moved = d > 0.25
            n_moved = int(moved.sum())
            print(f"\nthresholding {which} intensity_scale={iscale}: {n_moved} bins differ by one pixel's weight")
            assert n_moved <= 8
            assert float(d[~moved].max()) < 0.05
            if n_moved:
                if iscale:
                    assert float(d[moved].max()) <= (3.0 + 1e-6) ** 0.5 * 1.05
                else:
                    assert float((d[moved] - 1.0).abs().max()) < 0.05          # whole counts
        else:
            assert rel_err(h, ref) < 1e-4


def test_hist_exact_integer_counts():
    """Bit-exact bin COUNTS (thresholding, intensity_scale=False) on pixels whose log-chroma is
    provably away from every bin edge, so the result does not depend on the last ulp of log()."""
    from style_transfer_amd.losses import RGBuvHistBlock
    g = torch.Generator().manual_seed(123)
    cand = torch.rand(400000, 3, generator=g, dtype=torch.float64) * 0.98 + 0.01
    logc = torch.log(cand + 1e-6)
    bins = torch.tensor(np.linspace(-3, 3, 64))
    keep = torch.ones(cand.shape[0], dtype=torch.bool)
    thr = 6.0 / 64 / 2
    for a in range(3):
        for b in range(3):
            if a != b:
                d = (logc[:, a] - logc[:, b]).unsqueeze(1) - bins.unsqueeze(0)
                keep &= ((d.abs() - thr).abs() > 1e-4).all(dim=1)
    px = cand[keep][: 2 * 100 * 100].float()
    assert px.shape[0] == 20000
    img = px.t().reshape(3, 2, 100, 100).permute(1, 0, 2, 3).contiguous()      # [2,3,100,100] (<= insz: no resize)
    ref = R.rgbuv_hist(img, method="thresholding", intensity_scale=False)
    blk = RGBuvHistBlock(insz=128, h=64, intensity_scale=False, method="thresholding", hist_boundary=[-3, 3])
    h = blk(img.cuda()).cpu()
    # un-normalise with the exact integer total of an independent direct count
    raw_ref = _raw_counts(img)
    tot = raw_ref.sum(dim=(1, 2, 3), keepdim=True)
    assert float(tot.min()) > 1000
    raw_hip = torch.round(h.double() * (tot.double() + 1e-6))
    assert torch.equal(raw_hip, raw_ref.double())     # bit-exact integer bin counts
    assert torch.equal(h, ref)                        # and identical normalised fp32 values vs the oracle


def _raw_counts(img, iscale=False):
    """Bin counts (x Iy when ``iscale``) by direct counting (independent of both implementations' float paths)."""
    logc = torch.log(img.clamp(0, 1) + 1e-6)             # fp32, as both implementations do
    bins = torch.tensor(np.linspace(-3, 3, 64))
    thr = 6.0 / 64 / 2
    out = torch.zeros(img.shape[0], 3, 64, 64)
    for l in range(img.shape[0]):
        L = logc[l].reshape(3, -1)
        for ch, (o1, o2) in enumerate(((1, 2), (0, 2), (0, 1))):
            u = (L[ch] - L[o1]).double().unsqueeze(1)
            v = (L[ch] - L[o2]).double().unsqueeze(1)
            mu = ((u - bins).abs() <= thr).float()
            mv = ((v - bins).abs() <= thr).float()
            if iscale:
                mu = mu * torch.sqrt((img[l].clamp(0, 1).reshape(3, -1) ** 2).sum(dim=0) + 1e-6).unsqueeze(1)
            out[l, ch] = mu.t() @ mv
    return out


def test_hist_loss_golden(golden_dir):
    from style_transfer_amd.losses import HistLoss
    g = np.load(os.path.join(golden_dir, "hist.npz"))
    hl = HistLoss()
    x = torch.from_numpy(g["small"]).cuda().requires_grad_(True)
    v = hl(x, torch.from_numpy(g["tgt_small"]).cuda())
    v.backward()
    assert float(v) == pytest.approx(float(g["histloss"]), rel=1e-4)
    assert rel_err(x.grad, torch.from_numpy(g["histloss_dsmall"])) < 2e-3
    xb = torch.from_numpy(g["big"]).cuda().requires_grad_(True)
    tgt = torch.rand(2, 3, 150, 141, generator=torch.Generator().manual_seed(41)).cuda()
    v = hl(xb, tgt)
    v.backward()
    assert float(v) == pytest.approx(float(g["histloss_big"]), rel=1e-4)
    assert rel_err(xb.grad, torch.from_numpy(g["histloss_dbig"])) < 2e-3


def test_hist_loss_known_answer(golden_dir):
    from style_transfer_amd.losses import HistLoss
    kat = np.load(os.path.join(golden_dir, "kat.npz"))
    torch.manual_seed(0)
    img = torch.rand(8, 3, 256, 256)
    tgt = torch.rand(8, 3, 256, 256)
    v = HistLoss()(img.cuda(), tgt.cuda())
    assert float(v) == pytest.approx(float(kat["histloss"]), rel=1e-4)       # 2.91424...


def rel_l2(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


HIST_OPTION_CASES = {
    "h32_iq": dict(h=32, method="inverse-quadratic"), "h32_rbf": dict(h=32, method="RBF"),
    "h32_thr": dict(h=32, method="thresholding", intensity_scale=False),
    "h48_iq": dict(h=48, method="inverse-quadratic", hist_boundary=[-2, 2.5]),
    "green_iq": dict(h=64, method="inverse-quadratic", green_only=True),
    "green_thr16": dict(h=16, method="thresholding", green_only=True),
    "samp_iq": dict(h=64, method="inverse-quadratic", resizing="sampling"),
    "samp_green32": dict(h=32, method="RBF", resizing="sampling", green_only=True),
}


@pytest.mark.parametrize("tag", sorted(HIST_OPTION_CASES))
def test_hist_block_options_against_reference_golden(golden_dir, tag):
    """The settings HistLoss never uses -- h < 64, green_only, resizing='sampling' (losses.py:175-183, 195-196,
    278-281) -- against the reference module's outputs (tests/golden/hist_options.npz)."""
    from style_transfer_amd.losses import RGBuvHistBlock
    g = np.load(os.path.join(golden_dir, "hist_options.npz"))
    kw = dict(HIST_OPTION_CASES[tag])
    if "hist_boundary" in kw:
        kw["hist_boundary"] = list(kw["hist_boundary"])
    blk = RGBuvHistBlock(insz=128, **kw)
    for which in ("small", "big"):
        with torch.no_grad():
            got = blk(torch.from_numpy(g[which]).cuda()).cpu()
        want = torch.from_numpy(g[f"{tag}_{which}"])
        assert got.shape == want.shape
        d = (got - want).abs()
        if "thr" in tag:        # a pixel on a bin edge may fall either way: few bins, single-pixel moves
            assert d.max() < 2e-2 and (d > 1e-6).sum() <= 8, (tag, which, float(d.max()), int((d > 1e-6).sum()))
        else:
            assert d.max() <= 1e-4 * want.abs().max() + 1e-7, (tag, which, float(d.max()))
        torch.testing.assert_close(got.sum(dim=(1, 2, 3)), want.sum(dim=(1, 2, 3)), rtol=1e-5, atol=1e-6)


def test_hist_block_options_gradient_against_reference_golden(golden_dir):
    from style_transfer_amd.losses import RGBuvHistBlock
    g = np.load(os.path.join(golden_dir, "hist_options.npz"))
    blk = RGBuvHistBlock(insz=128, h=32, method="inverse-quadratic", green_only=True, resizing="sampling")
    x = torch.from_numpy(g["big"]).cuda().requires_grad_(True)
    (blk(x) * torch.from_numpy(g["grad_w"]).cuda()).sum().backward()
    want = torch.from_numpy(g["grad_dbig"])
    assert rel_l2(x.grad.cpu(), want) < 2e-3
    assert ((x.grad.cpu() != 0) == (want != 0)).all()          # only the sampled pixels receive a gradient


def test_hist_block_rejects_more_than_64_bins():
    from style_transfer_amd.losses import RGBuvHistBlock
    with pytest.raises(NotImplementedError):
        RGBuvHistBlock(h=128)
