#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules.

Run in the build container only (the reference checkout never travels):

    cd /root/repo && python oracle/gen_golden.py [--ref /root/reference]

Recipe: cwd = the reference root (its sources do
``sys.path.insert(0, './')``), bytecode writes off (the tree is read-only),
``torchvision`` replaced by an empty module object because generator.py /
losses.py import ``torchvision.transforms`` at module top but only use it
under ``if __name__ == "__main__"``.  ``Encoder()`` hard-loads a weight blob
that is absent from the checkout (.MISSING_LARGE_BLOBS), so ``torch.load`` is
pointed at a seeded synthetic state dict while the module is constructed.

Only inputs (or the seeds that make them) and the reference's outputs are
stored -- data, not source.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

import torch  # noqa: E402

from oracle import ref_ops as R  # noqa: E402  (only for the seeded weight/input makers)


def _import_reference(ref_root: str):
    os.chdir(ref_root)
    sys.path.insert(0, "./")
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    import src.models.generator as G
    import src.models.losses as L
    import src.models.transformer as T
    return G, T, L


def _np(t):
    return t.detach().cpu().numpy()


def hist_options(L, out_dir):
    """RGBuvHistBlock with the settings HistLoss never uses (losses.py:175-183 'sampling', :195-196 / :278-281
    green_only, h != 64): outputs of the reference's module on seeded inputs."""
    ho = {}
    gh = torch.Generator().manual_seed(44)
    small = torch.rand(2, 3, 24, 20, generator=gh) * 1.4 - 0.2
    big = torch.rand(2, 3, 150, 141, generator=gh)
    ho["small"], ho["big"] = _np(small), _np(big)
    cases = [("h32_iq", dict(h=32, method="inverse-quadratic")), ("h32_rbf", dict(h=32, method="RBF")),
             ("h32_thr", dict(h=32, method="thresholding", intensity_scale=False)),
             ("h48_iq", dict(h=48, method="inverse-quadratic", hist_boundary=[-2, 2.5])),
             ("green_iq", dict(h=64, method="inverse-quadratic", green_only=True)),
             ("green_thr16", dict(h=16, method="thresholding", green_only=True)),
             ("samp_iq", dict(h=64, method="inverse-quadratic", resizing="sampling")),
             ("samp_green32", dict(h=32, method="RBF", resizing="sampling", green_only=True))]
    for tag, kw in cases:
        blk = L.RGBuvHistBlock(insz=128, device="cpu", **kw)
        with torch.no_grad():
            ho[f"{tag}_small"] = _np(blk(small))
            ho[f"{tag}_big"] = _np(blk(big))
    # gradient through one of them (autograd of the reference)
    blk = L.RGBuvHistBlock(insz=128, device="cpu", h=32, method="inverse-quadratic", green_only=True, resizing="sampling")
    wgt = torch.rand(2, 1, 32, 32, generator=gh)
    xb = big.clone().requires_grad_(True)
    (blk(xb) * wgt).sum().backward()
    ho["grad_w"], ho["grad_dbig"] = _np(wgt), _np(xb.grad)
    np.savez_compressed(os.path.join(out_dir, "hist_options.npz"), **ho)
    print("hist_options.npz", os.path.getsize(os.path.join(out_dir, "hist_options.npz")), "bytes")


def custom_transforms(ref_root, out_dir):
    """The reference's two custom transforms (src/utils/custom_transform.py:9-46) run from its own module.  The module
    imports ``torchvision.transforms`` (absent from this image) at the top and calls exactly one torchvision function,
    ``TF.to_tensor`` on the uint8 HWC array it has just built (:26): the stand-in module's ``to_tensor`` RECORDS that
    array (the last value the reference itself computes) and returns the documented uint8 -> [0, 1] CHW tensor.
    Stored: inputs, numpy seeds, the recorded uint8 arrays, the pass-through cases, and RGBToGrayscaleStacked outputs
    (blue weight 0.144 as written at :41).  torchvision's own ops are NOT stood in for: they stay parity-unpinned."""
    os.chdir(ref_root)
    sys.path.insert(0, "./")
    captured = []

    def to_tensor(arr):
        captured.append(np.array(arr, copy=True))
        return torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).to(torch.float32).div(255)

    tv = sys.modules.get("torchvision") or types.ModuleType("torchvision")
    tvt = sys.modules.get("torchvision.transforms") or types.ModuleType("torchvision.transforms")
    tvtt = types.ModuleType("torchvision.transforms.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvf.to_tensor = to_tensor
    tv.transforms, tvt.transforms, tvt.functional = tvt, tvtt, tvf
    for name, mod in (("torchvision", tv), ("torchvision.transforms", tvt), ("torchvision.transforms.transforms", tvtt),
                      ("torchvision.transforms.functional", tvf)):
        sys.modules[name] = mod
    import matplotlib
    matplotlib.use("Agg")
    import src.utils.custom_transform as CT

    out = {}
    g = torch.Generator().manual_seed(77)
    imgs = {"a": torch.rand(3, 24, 20, generator=g), "b": torch.rand(3, 37, 45, generator=g) * 0.6 + 0.2,
            "c": torch.rand(1, 3, 16, 16, generator=g)}          # c: the leading 1 the reference squeezes away (:18)
    for k, v in imgs.items():
        out[f"img_{k}"] = _np(v)
    # RGBToGrayscaleStacked(enable) -- :32-46
    for k in ("a", "b"):
        out[f"gray_on_{k}"] = _np(CT.RGBToGrayscaleStacked(True)(imgs[k].clone()))
        out[f"gray_off_{k}"] = _np(CT.RGBToGrayscaleStacked(False)(imgs[k].clone()))
    # AddGaussianNoise -- :9-29, train.py:131's settings and the class defaults; seeds chosen so that both the noisy
    # and the pass-through branch occur
    cases = []
    for k, kw, seeds in (("a", dict(mean=0.5, sigma_range=(0., 0.08), p=0.5), (0, 1, 2, 3, 4, 5)),
                         ("b", dict(mean=0.5, sigma_range=(0., 0.08), p=0.9), (10, 11)),
                         ("c", dict(), (20, 21, 22, 23))):
        t = CT.AddGaussianNoise(**kw)
        for seed in seeds:
            np.random.seed(seed)
            captured.clear()
            res = t(imgs[k].clone())
            tag = f"noise_{k}_{seed}"
            out[f"{tag}_applied"] = np.bool_(len(captured) == 1)
            if captured:
                assert captured[0].dtype == np.uint8
                out[f"{tag}_u8"] = captured[0]
            else:
                assert torch.equal(res, imgs[k])
            cases.append((k, seed, kw.get("mean", 0.0), *kw.get("sigma_range", (0.0, 0.06)), kw.get("p", 0.5)))
    out["noise_cases"] = np.array([(ord(k), s, m, lo, hi, p) for k, s, m, lo, hi, p in cases], dtype=np.float64)
    path = os.path.join(out_dir, "transforms.npz")
    np.savez_compressed(path, **out)
    n_app = sum(bool(out[f"noise_{k}_{s}_applied"]) for k, s, *_ in cases)
    print("transforms.npz", os.path.getsize(path), "bytes;", n_app, "of", len(cases), "noise cases applied")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default=None, choices=[None, "hist_options", "transforms"],
                    help="write just this file (hist_options: round 2, transforms: round 3; the others stay as "
                         "generated in round 1)")
    args = ap.parse_args()
    out_dir = os.path.abspath(args.out)
    os.makedirs(out_dir, exist_ok=True)
    if args.only == "transforms":
        custom_transforms(args.ref, out_dir)
        return
    G, T, L = _import_reference(args.ref)
    torch.set_num_threads(8)
    if args.only == "hist_options":
        hist_options(L, out_dir)
        return

    # ---- 1. SymmetricPadding2D ------------------------------------------------
    x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    np.savez_compressed(
        os.path.join(out_dir, "sympad.npz"),
        x=_np(x),
        pad1=_np(G.SymmetricPadding2D((1, 1, 1, 1))(x)),
        pad2=_np(G.SymmetricPadding2D((2, 2, 2, 2))(x)),
        pad_asym=_np(G.SymmetricPadding2D((1, 3, 0, 6))(x)),
    )

    # ---- 2. Encoder / Decoder --------------------------------------------------
    enc_sd = R.make_encoder_state(seed=0)
    dec_sd = R.make_decoder_state(seed=1)
    real_load = torch.load
    torch.load = lambda *a, **k: enc_sd
    try:
        enc = G.Encoder().eval()
    finally:
        torch.load = real_load
    dec = G.Decoder().eval()
    dec.load_state_dict(dec_sd)
    assert list(dec.state_dict().keys()) == list(dec_sd.keys())
    gen = {}
    for name, shape, seed in (("a", (1, 3, 32, 32), 10), ("b", (2, 3, 37, 45), 11)):
        xin = torch.rand(*shape, generator=torch.Generator().manual_seed(seed))
        with torch.no_grad():
            f = enc(xin)
            y = dec(f)
        gen[f"{name}_seed"] = np.int64(seed)
        gen[f"{name}_shape"] = np.array(shape, dtype=np.int64)
        gen[f"{name}_feat"] = _np(f)
        gen[f"{name}_dec"] = _np(y)
    # decoder gradient wrt its input and two of its weights (training needs dgrad+wgrad)
    fin = torch.randn(1, 512, 3, 5, generator=torch.Generator().manual_seed(12)).requires_grad_(True)
    dec.zero_grad()
    yo = dec(fin)
    gout = torch.randn(yo.shape, generator=torch.Generator().manual_seed(13))
    (yo * gout).sum().backward()
    gen["g_out"] = _np(yo)
    gen["g_din"] = _np(fin.grad)
    gen["g_dw28"] = _np(dec.decoder[28].weight.grad)
    gen["g_db28"] = _np(dec.decoder[28].bias.grad)
    gen["g_dw1_slice"] = _np(dec.decoder[1].weight.grad[:8, :16])
    gen["g_db1"] = _np(dec.decoder[1].bias.grad)
    gen["g_dw18_slice"] = _np(dec.decoder[18].weight.grad[:8, :16])
    np.savez_compressed(os.path.join(out_dir, "generator.npz"), **gen)

    # ---- 3. AdaIN ---------------------------------------------------------------
    ad = {}
    gc = torch.Generator().manual_seed(20)
    content = torch.randn(2, 6, 7, 9, generator=gc) * 2.0 + 0.5
    style = torch.randn(2, 6, 5, 4, generator=gc) * 0.7 - 1.0
    gup = torch.randn(2, 6, 7, 9, generator=gc)
    ad["content"], ad["style"], ad["gup"] = _np(content), _np(style), _np(gup)
    for tag, lim in (("1e-5", 1e-5), ("0.1", 0.1), ("0.3", 0.3)):
        m = T.AdaIN(eps=lim)
        with torch.no_grad():
            m.content_eps.fill_(4e-4)      # sqrt = 0.02  -> inside (1e-5, 0.1)
            m.style_eps.fill_(2.5e-3)      # sqrt = 0.05
        c = content.clone().requires_grad_(True)
        s = style.clone().requires_grad_(True)
        o = m(c, s)
        (o * gup).sum().backward()
        ad[f"out_{tag}"] = _np(o)
        ad[f"dcontent_{tag}"] = _np(c.grad)
        ad[f"dstyle_{tag}"] = _np(s.grad)
        ad[f"dceps_{tag}"] = _np(m.content_eps.grad) if m.content_eps.grad is not None else np.zeros(1, np.float32)
        ad[f"dseps_{tag}"] = _np(m.style_eps.grad) if m.style_eps.grad is not None else np.zeros(1, np.float32)
        eps_now = m.get_current_eps()
        ad[f"eps_{tag}"] = np.array([float(e) for e in eps_now], dtype=np.float64)
    mean, std = T.AdaIN.calc_mean_std(content, eps=1e-5)
    ad["mean"], ad["std"] = _np(mean), _np(std)
    # adjust_sigmoid branches
    ad["sig_lo"] = np.float64(float(T.AdaIN.adjust_sigmoid(torch.tensor([1e-12]), 0.3)))
    ad["sig_hi"] = np.float64(float(T.AdaIN.adjust_sigmoid(torch.tensor([0.5]), 0.3)))
    ad["sig_mid"] = np.float64(float(T.AdaIN.adjust_sigmoid(torch.tensor([-0.01]), 0.3)))
    np.savez_compressed(os.path.join(out_dir, "adain.npz"), **ad)

    # ---- 4. Losses ---------------------------------------------------------------
    ls = {}
    gl = torch.Generator().manual_seed(30)
    a = torch.randn(2, 8, 6, 5, generator=gl)
    b = torch.randn(2, 8, 6, 5, generator=gl) * 1.5 + 0.3
    ls["a"], ls["b"] = _np(a), _np(b)

    def fwd_grad(fn, x, *rest):
        xx = x.clone().requires_grad_(True)
        v = fn(xx, *rest)
        v.backward()
        return _np(v), _np(xx.grad)

    ls["adain_style"], ls["adain_style_da"] = fwd_grad(L.AdaINStyleLoss(), a, b)
    ls["content"], ls["content_da"] = fwd_grad(L.AdaINContentLoss(), a, b)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vin = L.VincentStyleLoss()
    av = torch.randn(2, 8, 9, 7, generator=gl)
    bv = torch.randn(2, 8, 9, 7, generator=gl) * 1.3
    ls["av"], ls["bv"] = _np(av), _np(bv)
    ls["vincent"], ls["vincent_da"] = fwd_grad(vin, av, bv)
    mu, cov = L.VincentStyleLoss.calc_2_moments(av)
    ls["vincent_mu"], ls["vincent_cov"] = _np(mu), _np(cov)
    img = torch.rand(2, 3, 11, 13, generator=gl) * 3 - 1
    ls["img"] = _np(img)
    ls["tv"], ls["tv_dimg"] = fwd_grad(L.TVLoss(), img)
    np.savez_compressed(os.path.join(out_dir, "losses.npz"), **ls)

    # ---- 5. Histogram -------------------------------------------------------------
    hs = {}
    gh = torch.Generator().manual_seed(40)
    small = torch.rand(2, 3, 24, 20, generator=gh) * 1.4 - 0.2     # <= insz, exercises clamp
    big = torch.rand(2, 3, 150, 141, generator=gh)                  # > insz -> bilinear to 128x128
    tgt_small = torch.rand(2, 3, 24, 20, generator=gh)
    hs["small"], hs["big"], hs["tgt_small"] = _np(small), _np(big), _np(tgt_small)
    for method in ("inverse-quadratic", "RBF", "thresholding"):
        for iscale in (True, False):
            blk = L.RGBuvHistBlock(insz=128, h=64, intensity_scale=iscale, method=method,
                                   hist_boundary=[-3, 3], device="cpu")
            tag = f"{method}_{int(iscale)}"
            with torch.no_grad():
                hs[f"hist_small_{tag}"] = _np(blk(small))
                hs[f"hist_big_{tag}"] = _np(blk(big))
    hl = L.HistLoss()
    hl.histogram_block.device = "cpu"
    hs["histloss"], hs["histloss_dsmall"] = fwd_grad(hl, small, tgt_small)
    bigc = big.clone().requires_grad_(True)
    v = hl(bigc, torch.rand(2, 3, 150, 141, generator=torch.Generator().manual_seed(41)))
    v.backward()
    hs["histloss_big"], hs["histloss_dbig"] = _np(v), _np(bigc.grad)
    # unnormalised integer counts (thresholding, intensity_scale False): exact integers
    blk = L.RGBuvHistBlock(insz=128, h=64, intensity_scale=False, method="thresholding",
                           hist_boundary=[-3, 3], device="cpu")
    with torch.no_grad():
        hs["hist_counts_norm"] = _np(blk(small))
    np.savez_compressed(os.path.join(out_dir, "hist.npz"), **hs)

    # ---- 6. Known-answer tests ----------------------------
    torch.manual_seed(0)
    img = torch.rand(8, 3, 256, 256)
    tv = float(L.TVLoss()(img))
    torch.manual_seed(0)
    img = torch.rand(8, 3, 256, 256)
    tgt = torch.rand(8, 3, 256, 256)
    with torch.no_grad():
        hv = float(hl(img, tgt))
    np.savez_compressed(os.path.join(out_dir, "kat.npz"), tv=np.float64(tv), histloss=np.float64(hv))
    print("TV KAT", tv, "HistLoss KAT", hv)
    print("golden vectors written to", out_dir)
    for f in sorted(os.listdir(out_dir)):
        print(f"  {f:20s} {os.path.getsize(os.path.join(out_dir, f))} bytes")


if __name__ == "__main__":
    main()
