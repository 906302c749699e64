"""GPU-side augmentation (csrc/augment.hip, gpu_augment.py) against the CPU restatement of the reference's
post-ToTensor transforms (transforms.py; src/train.py:116-135, src/utils/custom_transform.py:9-46) on identical,
explicit per-image parameters.  PARITY UNPINNED against torchvision itself (0.15.1 is not installed here): the CPU
restatement follows its documented behaviour; the two custom transforms are pinned by vectors generated from the
reference's own module (tests/golden/transforms.npz, oracle/gen_golden.py --only transforms)."""
import math

import numpy as np
import pytest
import torch

from style_transfer_amd import _lib
from style_transfer_amd import transforms as T
from style_transfer_amd.gpu_augment import GpuAugment

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def reference_apply(img, i, params, aug, noise_field=None):
    """The transforms of a GpuAugment on ONE image [3, h, w] with the CPU restatement (transforms.py) and the explicit
    draws ``params``: the checker of the whole-pipeline test."""
    img = img.cpu()
    if aug.grayscale:
        img = T.RGBToGrayscaleStacked(True)(img)
    if aug.rotation_degrees:
        img = T.rotate(img, float(params["angle_deg"][i]))
    if aug.noise is not None and float(params["noise_sigma"][i]) >= 0:
        img = T.add_noise(img, noise_field[i].permute(1, 2, 0).double().cpu().numpy())
    if float(params["sharp_factor"][i]) >= 0:
        img = T.adjust_sharpness(img, float(params["sharp_factor"][i]))
    jp = params["jitter_param"][i]
    val = lambda k: None if math.isnan(float(jp[k])) else float(jp[k])            # noqa: E731
    img = T.color_jitter(img, [int(v) for v in params["jitter_order"][i]], val(0), val(1), val(2), val(3))
    if int(params["flip"][i]):
        img = img.flip(-1)
    return img


def batch(n, h, w, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 3, h, w, generator=g)
    x[0, :, : h // 2] = (x[0, :, : h // 2] * 4).floor() / 4          # flat regions: grey pixels, ties in max/min
    if n > 1:
        x[1] = x[1].mean(0, keepdim=True)                            # a grey image (maxc == minc everywhere)
    return x


def f32(v, n):
    return torch.full((n,), float(v))


def call(name, *args):
    lib = _lib.load()
    _lib.check(getattr(lib, name)(*args, _lib.stream_ptr()), name)


def P(t, dtype=torch.float32):
    return _lib.dev_ptr(t, "t", dtype)


@pytest.mark.parametrize("shape", [(3, 64, 64), (2, 37, 53), (8, 256, 256)])
def test_gray_stack(shape):
    n, h, w = shape
    x = batch(n, h, w)
    xd, yd = x.to(DEV), torch.empty(n, 3, h, w, device=DEV)
    call("st_aug_gray_stack", P(xd), P(yd), n, h * w)
    want = torch.stack([T.RGBToGrayscaleStacked(True)(x[i]) for i in range(n)])
    torch.testing.assert_close(yd.cpu(), want, rtol=0, atol=2e-7)


@pytest.mark.parametrize("shape", [(4, 64, 64), (3, 37, 53), (8, 256, 256)])
def test_rotate_nearest_same_canvas(shape):
    n, h, w = shape
    x = batch(n, h, w, 1)
    ang = torch.tensor(([0.0, 90.0, -90.0, 33.3, -71.9, 12.0, 45.0, -1.5] * 2)[:n])
    xd, yd, ad = x.to(DEV), torch.empty(n, 3, h, w, device=DEV), torch.deg2rad(ang).to(DEV)
    call("st_aug_rotate", P(xd), P(yd), P(ad), n, 3, h, w)
    got = yd.cpu()
    for i in range(n):
        want = T.rotate(x[i], float(ang[i]))
        # nearest sampling: a source coordinate that falls within fp32 rounding of a pixel boundary may pick the
        # neighbour; everything else is a bit-exact copy
        wrong = (got[i] != want).any(0).float().mean().item()
        assert wrong <= 4e-3, (i, float(ang[i]), wrong)
    torch.testing.assert_close(got[0], x[0], rtol=0, atol=0)        # angle 0 = identity


@pytest.mark.parametrize("shape", [(4, 32, 48), (8, 256, 256)])
def test_gaussian_noise_matches_the_uint8_round_trip(shape):
    n, h, w = shape
    x = batch(n, h, w, 2)
    g = torch.Generator().manual_seed(5)
    sigma = torch.tensor(([0.05, -1.0, 0.0, 0.08] * 2)[:n])
    noise = torch.randn(n, 3, h, w, generator=g) * sigma.clamp_min(0).view(n, 1, 1, 1) + 0.5
    xd, yd, ws = x.to(DEV), torch.empty(n, 3, h, w, device=DEV), torch.empty(2 * n, device=DEV)
    nd, sd = noise.to(DEV), sigma.to(DEV)         # named: a temporary's block would be handed to the next allocation
    call("st_aug_gaussian_noise", P(xd), P(nd), P(sd), P(yd), P(ws), n, 3 * h * w)
    got = yd.cpu()
    for i in range(n):
        if sigma[i] < 0:
            torch.testing.assert_close(got[i], x[i], rtol=0, atol=0)
            continue
        want = T.add_noise(x[i], noise[i].permute(1, 2, 0).double().numpy())
        d = (got[i] - want).abs()
        # the checker interpolates in float64, the kernel in fp32: truncation to uint8 can land one level apart
        assert d.max().item() <= 1.0 / 255 + 1e-6
        assert (d > 1e-6).float().mean().item() < 0.02
        levels = got[i] * 255
        assert (levels - levels.round()).abs().max().item() < 1e-3      # every value is a uint8 level


def _golden_transforms():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transforms.npz"))


def test_custom_transforms_match_reference_generated_vectors():
    """The two transforms the reference wrote itself (src/utils/custom_transform.py:9-46), HIP kernels against vectors
    produced by the reference's own module (oracle/gen_golden.py --only transforms): RGBToGrayscaleStacked outputs and
    the uint8 array AddGaussianNoise hands to ToTensor, for the numpy RNG state the fixture names."""
    gold = _golden_transforms()
    for k in ("a", "b"):
        x = torch.from_numpy(gold[f"img_{k}"])
        _, h, w = x.shape
        xd, yd = x.unsqueeze(0).to(DEV), torch.empty(1, 3, h, w, device=DEV)
        call("st_aug_gray_stack", P(xd), P(yd), 1, h * w)
        torch.testing.assert_close(yd[0].cpu(), torch.from_numpy(gold[f"gray_on_{k}"]), rtol=0, atol=2e-7)
    seen = 0
    for key, seed, mean, lo, hi, p in gold["noise_cases"]:
        k, seed = chr(int(key)), int(seed)
        if not bool(gold[f"noise_{k}_{seed}_applied"]):
            continue
        x = torch.from_numpy(gold[f"img_{k}"]).reshape(3, *gold[f"img_{k}"].shape[-2:])
        _, h, w = x.shape
        np.random.seed(seed)                       # the reference's draw order: gate, sigma, field (:16, :22, :23)
        assert np.random.rand() < p
        sigma = np.random.uniform(lo, hi)
        field = np.random.normal(mean, sigma, (h, w, 3))
        nd = torch.from_numpy(field).permute(2, 0, 1).float().unsqueeze(0).contiguous().to(DEV)
        xd, yd, ws = x.unsqueeze(0).to(DEV), torch.empty(1, 3, h, w, device=DEV), torch.empty(2, device=DEV)
        sd = torch.tensor([sigma], dtype=torch.float32, device=DEV)
        call("st_aug_gaussian_noise", P(xd), P(nd), P(sd), P(yd), P(ws), 1, 3 * h * w)
        got = (yd[0].cpu() * 255).round().permute(1, 2, 0).numpy().astype(np.int64)
        want = gold[f"noise_{k}_{seed}_u8"].astype(np.int64)
        d = np.abs(got - want)
        # the reference interpolates in float64 (np.interp), the kernel in fp32 on an fp32 copy of the field:
        # truncation to uint8 can land one level apart on a few values
        assert d.max() <= 1 and (d > 0).mean() < 0.02, (k, seed, d.max(), (d > 0).mean())
        seen += 1
    assert seen >= 5


@pytest.mark.parametrize("shape", [(4, 64, 64), (3, 37, 53), (2, 2, 9), (8, 256, 256)])
def test_sharpness(shape):
    n, h, w = shape
    x = batch(n, h, w, 3)
    fac = torch.tensor(([1.5, -1.0, 0.0, 2.0] * 2)[:n])
    xd, yd, fd = x.to(DEV), torch.empty(n, 3, h, w, device=DEV), fac.to(DEV)
    call("st_aug_sharpness", P(xd), P(yd), P(fd), n, 3, h, w)
    for i in range(n):
        want = x[i] if fac[i] < 0 else T.adjust_sharpness(x[i], float(fac[i]))
        torch.testing.assert_close(yd[i].cpu(), want, rtol=0, atol=1e-6)


@pytest.mark.parametrize("code,param", [(0, 0.8), (0, 1.2), (1, 0.6), (1, 1.4), (2, 0.7), (2, 1.3), (3, -0.1), (3, 0.07),
                                        (-1, 0.0)])
@pytest.mark.parametrize("shape", [(3, 48, 40), (8, 256, 256)])
def test_color_jitter_slot(code, param, shape):
    n, h, w = shape
    x = batch(n, h, w, 4)
    xd, yd, ws = x.to(DEV), torch.empty(n, 3, h, w, device=DEV), torch.empty(2 * n, device=DEV)
    cd, pd = torch.full((n,), code, dtype=torch.int32, device=DEV), f32(param, n).to(DEV)
    call("st_aug_color_jitter_slot", P(xd), P(yd), P(cd, torch.int32), P(pd), P(ws), n, h * w)
    kw = {0: "b", 1: "c", 2: "s", 3: "h"}
    for i in range(n):
        want = x[i] if code < 0 else T.color_jitter(x[i], [code], **{kw[code]: param})
        torch.testing.assert_close(yd[i].cpu(), want, rtol=0, atol=3e-6)


def test_hflip():
    n, h, w = 5, 19, 31
    x = batch(n, h, w, 6)
    fl = torch.tensor([1, 0, 1, 1, 0], dtype=torch.int32)
    xd, yd, fd = x.to(DEV), torch.empty(n, 3, h, w, device=DEV), fl.to(DEV)
    call("st_aug_hflip", P(xd), P(yd), P(fd, torch.int32), n, 3, h, w)
    want = torch.stack([x[i].flip(-1) if fl[i] else x[i] for i in range(n)])
    torch.testing.assert_close(yd.cpu(), want, rtol=0, atol=0)


def test_bad_arguments_are_refused():
    lib = _lib.load()
    x, z = torch.zeros(1, 3, 4, 4, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    assert lib.st_aug_hflip(P(x), P(x), P(z, torch.int32), 1, 3, 4, 4, _lib.stream_ptr()) != 0      # in place
    assert b"in-place" in lib.st_last_error()
    assert lib.st_aug_gray_stack(P(x), P(x), 0, 16, _lib.stream_ptr()) != 0
    with pytest.raises(_lib.StHipError):
        GpuAugment()(torch.zeros(1, 3, 4, 4))                          # a CPU batch: no fallback


@pytest.mark.parametrize("which", ["content", "content_gray", "style"])
def test_whole_pipeline_against_the_cpu_transforms(which):
    n, h, w = 8, 96, 80
    aug = {"content": GpuAugment.for_content(seed=11), "content_gray": GpuAugment.for_content(grayscale=True, seed=12),
           "style": GpuAugment.for_style(seed=13)}[which]
    x = batch(n, h, w, 7)
    params = aug.draw(n)
    noise = None
    if aug.noise is not None:
        params["noise_sigma"][0], params["noise_sigma"][1] = 0.05, -1.0           # both branches present
        g = torch.Generator().manual_seed(9)
        noise = torch.randn(n, 3, h, w, generator=g) * params["noise_sigma"].clamp_min(0).view(n, 1, 1, 1) + 0.5
    before = x.clone()
    xd = x.to(DEV)
    got = aug(xd, params=params, noise_field=noise).cpu()
    torch.testing.assert_close(xd.cpu(), before, rtol=0, atol=0)                  # the caller's batch is only read
    assert got.shape == x.shape and got.min() >= 0 and got.max() <= 1
    for i in range(n):
        want = reference_apply(x[i], i, params, aug, noise)
        d = (got[i] - want).abs()
        # pixels the rotation picked differently (fp32 boundary cases) spread through the 3x3 blur and, for the
        # style path, through the uint8 truncation: bound their share, and the rest tightly
        assert (d > 1.5 / 255).float().mean().item() < 0.02, (which, i, (d > 1.5 / 255).float().mean().item())
        assert d.mean().item() < 2e-3


def test_draws_are_reproducible_and_the_call_uses_them():
    x = batch(4, 32, 32, 8).to(DEV)
    a, b = GpuAugment.for_content(seed=3), GpuAugment.for_content(seed=3)
    torch.testing.assert_close(a(x), b(x), rtol=0, atol=0)
    assert not torch.equal(a(x), b.__class__.for_content(seed=4)(x))
    # the style pipeline's Gaussian noise field comes from the augmenter's own seeded device generator too
    s1, s2 = GpuAugment.for_style(seed=9), GpuAugment.for_style(seed=9)
    torch.manual_seed(123)
    y1 = s1(x)
    torch.manual_seed(456)
    torch.testing.assert_close(s2(x), y1, rtol=0, atol=0)


def test_trainer_applies_the_gpu_augmentation_to_every_batch(tmp_path, monkeypatch):
    """Trainer.train with gpu_augment=(content, style): the step sees aug(batch), drawn from the augmenters' own
    generators, and the epoch runs through (eager and HIP-graph replay)."""
    import warnings
    from argparse import Namespace
    import oracle.ref_ops as R
    from style_transfer_amd import trainer as trainer_mod
    from style_transfer_amd.trainer import Trainer

    class Loader(list):
        dataset = list(range(4))

    g = torch.Generator().manual_seed(3)
    batches = Loader([{"content_image": torch.rand(2, 3, 32, 32, generator=g),
                       "style_image": torch.rand(2, 3, 32, 32, generator=g)} for _ in range(2)])
    seen = []
    real = trainer_mod.Trainer.optimize_step

    def spy(self, step, opt, bucket, c, s):
        seen.append((c.clone(), s.clone()))
        return real(self, step, opt, bucket, c, s)
    monkeypatch.setattr(trainer_mod.Trainer, "optimize_step", spy)
    kw = dict(output_dir=str(tmp_path), num_train_epochs=1, per_device_batch_size=2, learning_rate=1e-4,
              alpha=1.0, beta=10.0, gamma=1.0, delta=10.0, eps=0.1, save_best=False, do_decoder_train=True,
              optim_name={"optim_name": "adam"}, content_layers_idx=[21], style_layers_idx=[2, 7, 14, 19, 25],
              config=Namespace(), encoder_weights=R.make_encoder_state(0), vgg_weights=R.make_loss_vgg_state(2, upto=25))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hist = Trainer({"train": batches}, gpu_augment=(GpuAugment.for_content(seed=5), GpuAugment.for_style(seed=6)),
                       **kw).train()
    assert len(hist) == 1 and math.isfinite(hist[0]["total"]) and len(seen) == 2
    twin_c, twin_s = GpuAugment.for_content(seed=5), GpuAugment.for_style(seed=6)
    torch.manual_seed(0)
    for (c, s), b in zip(seen, batches):
        torch.testing.assert_close(c, twin_c(b["content_image"].to(DEV)), rtol=0, atol=0)
        assert s.shape == b["style_image"].shape and not torch.equal(s.cpu(), b["style_image"])
        twin_s.draw(2)                                                  # keep the twin's generator in step
    monkeypatch.undo()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hist_g = Trainer({"train": batches}, use_hip_graph=True, device="cuda",
                         gpu_augment=(GpuAugment.for_content(seed=5), GpuAugment.for_style(seed=6)), **kw).train()
    assert len(hist_g) == 1 and math.isfinite(hist_g[0]["total"])
