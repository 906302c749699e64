"""GPU-side training augmentation: the transforms the reference composes AFTER ``ToTensor``
in src/train.py:116-135 -- content: RGBToGrayscaleStacked, RandomRotation(90), RandomAdjustSharpness(1.5),
ColorJitter, RandomHorizontalFlip; style: RandomRotation(90), AddGaussianNoise, RandomAdjustSharpness(1.5),
ColorJitter, RandomHorizontalFlip -- applied to a whole batch that is already in HBM by the HIP kernels of
csrc/augment.hip.  The DataLoader workers then only decode, resize and crop (PIL) -- the per-image tensor work, which
bounds the epoch once a training step takes 26 ms for 8 images, leaves the CPU.

Which image gets which transform, and every factor, is drawn per image on the HOST from one seeded
``torch.Generator`` (one small host-to-device copy per batch); the draws follow the same distributions as the
reference's transforms (probabilities, ranges, a uniformly random order of the four ColorJitter adjustments) but not
its RNG streams (torch global + numpy inside worker processes), which no implementation outside those processes can
reproduce.  ``params=`` replays explicit draws -- tests/test_gpu_augment.py compares every transform and both whole
pipelines with the CPU restatement (transforms.py) on identical parameters.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib


def _f(t):
    return _lib.dev_ptr(t, "tensor")


def _i(t):
    return _lib.dev_ptr(t, "tensor", torch.int32)


class GpuAugment:
    """``aug(batch)`` -> augmented batch [n, 3, h, w] (fp32, on the GPU).  Keyword defaults = src/train.py:116-135."""

    def __init__(self, grayscale: bool = False, rotation_degrees: Optional[float] = 90.0, noise: Optional[dict] = None,
                 sharpness_factor: Optional[float] = 1.5, sharpness_p: float = 0.5, brightness: float = 0.2,
                 contrast: float = 0.4, saturation: float = 0.3, hue: float = 0.1, flip_p: float = 0.5, seed: int = 0):
        self.grayscale, self.rotation_degrees, self.noise = grayscale, rotation_degrees, noise
        self.sharpness_factor, self.sharpness_p, self.flip_p = sharpness_factor, sharpness_p, flip_p
        self.jitter = {"brightness": brightness, "contrast": contrast, "saturation": saturation, "hue": hue}
        self.generator = torch.Generator().manual_seed(seed)
        self._noise_generators = {}       # device -> torch.Generator for the Gaussian noise field (seeded from `seed`)

    @classmethod
    def for_content(cls, grayscale: bool = False, seed: int = 0) -> "GpuAugment":
        return cls(grayscale=grayscale, contrast=0.4, seed=seed)                 # train.py:116-125

    @classmethod
    def for_style(cls, seed: int = 1) -> "GpuAugment":
        return cls(noise={"mean": 0.5, "sigma_range": (0.0, 0.08), "p": 0.5}, contrast=0.25, seed=seed)   # :126-135

    # ---- host-side draws -------------------------------------------------------------------------------------
    def draw(self, n: int) -> Dict[str, torch.Tensor]:
        g = self.generator
        u = lambda lo, hi: torch.empty(n).uniform_(lo, hi, generator=g)        # noqa: E731
        p = {}
        d = self.rotation_degrees
        p["angle_deg"] = u(-d, d) if d else torch.zeros(n)
        if self.noise is not None:
            on = torch.rand(n, generator=g) < self.noise["p"]
            sig = u(*self.noise["sigma_range"])
            p["noise_sigma"] = torch.where(on, sig, torch.full((n,), -1.0))
        on = torch.rand(n, generator=g) < self.sharpness_p if self.sharpness_factor is not None else torch.zeros(n, dtype=torch.bool)
        p["sharp_factor"] = torch.where(on, torch.full((n,), float(self.sharpness_factor or 0.0)), torch.full((n,), -1.0))
        p["jitter_order"] = torch.stack([torch.randperm(4, generator=g) for _ in range(n)]).to(torch.int32)
        j = self.jitter
        rng = lambda v: (max(0.0, 1 - v), 1 + v)                                 # noqa: E731
        p["jitter_param"] = torch.stack([
            u(*rng(j["brightness"])) if j["brightness"] else torch.full((n,), float("nan")),
            u(*rng(j["contrast"])) if j["contrast"] else torch.full((n,), float("nan")),
            u(*rng(j["saturation"])) if j["saturation"] else torch.full((n,), float("nan")),
            u(-j["hue"], j["hue"]) if j["hue"] else torch.full((n,), float("nan"))], dim=1)     # [n, 4] by adjustment
        p["flip"] = (torch.rand(n, generator=g) < self.flip_p).to(torch.int32)
        return p

    # ---- device side -------------------------------------------------------------------------------------------
    def __call__(self, x: torch.Tensor, params: Optional[Dict[str, torch.Tensor]] = None,
                 noise_field: Optional[torch.Tensor] = None) -> torch.Tensor:
        lib = _lib.load()
        assert x.dim() == 4 and x.shape[1] == 3, "GpuAugment expects [n, 3, h, w]"
        n, c, h, w = (int(v) for v in x.shape)
        p = params if params is not None else self.draw(n)
        dev, s = x.device, _lib.stream_ptr
        x = x.contiguous()
        # one host-to-device copy for all per-image parameters
        order, jp = p["jitter_order"].to(torch.int32), p["jitter_param"].float()
        slot_code = order.clone()
        slot_param = torch.gather(jp, 1, order.long())
        off = torch.isnan(slot_param)
        slot_code[off] = -1                                                     # adjustment switched off
        slot_param = torch.nan_to_num(slot_param, nan=0.0)
        fl = torch.cat([torch.deg2rad(p["angle_deg"].float()), p.get("noise_sigma", torch.full((n,), -1.0)).float(),
                        p["sharp_factor"].float(), slot_param.t().reshape(-1)]).to(dev, non_blocking=True)
        it = torch.cat([slot_code.t().reshape(-1), p["flip"].to(torch.int32)]).to(dev, non_blocking=True)
        angle, sigma, sharp, sparam = fl[:n], fl[n:2 * n], fl[2 * n:3 * n], fl[3 * n:].view(4, n)
        scode, flip = it[:4 * n].view(4, n), it[4 * n:]
        ws = torch.empty(2 * n, dtype=torch.float32, device=dev)
        bufs, state = (torch.empty_like(x), torch.empty_like(x)), {"cur": x, "k": 0}

        def step(name, call):                 # out-of-place ping-pong; the caller's batch is only ever read
            out = bufs[state["k"]]
            state["k"] ^= 1
            _lib.check(call(_f(state["cur"]), _f(out)), name)
            state["cur"] = out

        if self.grayscale:                                                      # train.py:120 (content only)
            step("st_aug_gray_stack", lambda a, b: lib.st_aug_gray_stack(a, b, n, h * w, s()))
        if self.rotation_degrees:
            step("st_aug_rotate", lambda a, b: lib.st_aug_rotate(a, b, _f(angle), n, c, h, w, s()))
        if self.noise is not None:                                              # train.py:131 (style only)
            if noise_field is None:
                gen = self._noise_generators.get(dev)
                if gen is None:
                    gen = self._noise_generators[dev] = torch.Generator(device=dev)
                    gen.manual_seed(self.generator.initial_seed() + 0x9E3779B1)
                std = sigma.clamp_min(0).view(n, 1, 1, 1)
                noise_field = torch.randn((n, c, h, w), device=dev, generator=gen) * std + float(self.noise["mean"])
            nf = noise_field.to(dev).contiguous()
            step("st_aug_gaussian_noise",
                 lambda a, b: lib.st_aug_gaussian_noise(a, _f(nf), _f(sigma), b, _f(ws), n, c * h * w, s()))
        if self.sharpness_factor is not None:
            step("st_aug_sharpness", lambda a, b: lib.st_aug_sharpness(a, b, _f(sharp), n, c, h, w, s()))
        for k in range(4):                                                      # ColorJitter: its four slots
            cd, pr = scode[k].contiguous(), sparam[k].contiguous()
            step("st_aug_color_jitter_slot",
                 lambda a, b: lib.st_aug_color_jitter_slot(a, b, _i(cd), _f(pr), _f(ws), n, h * w, s()))
        step("st_aug_hflip", lambda a, b: lib.st_aug_hflip(a, b, _i(flip), n, c, h, w, s()))
        return state["cur"]
