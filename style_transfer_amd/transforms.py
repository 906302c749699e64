"""torchvision-free restatement of the image transforms the reference's command lines compose
(src/train.py:116-135, src/inference.py:62-70, src/models/trainer.py:539-547, 686-689) and of its two
custom transforms (src/utils/custom_transform.py:9-46).

torchvision 0.15.1 (requirements.txt:3) is not installed in the build image, so the torchvision ops
are restated from their documented behaviour: PARITY UNPINNED for the random augmentations (their RNG
draw order follows torchvision's: torch's global generator).  ``Resize`` / ``ToTensor`` / ``Normalize``
are deterministic and exercised by tests/test_host_logic.py.  Host-side (CPU, PIL + torch): the data
pipeline is outside the HIP hot path.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


class Resize:
    """transforms.Resize on a PIL image: int -> shorter side to ``size`` keeping the aspect ratio
    (long side = int(size * long / short)), (h, w) -> exact; PIL bilinear (antialiased)."""

    def __init__(self, size: Union[int, Sequence[int]]):
        self.size = size

    def __call__(self, img: Image.Image) -> Image.Image:
        w, h = img.size
        if isinstance(self.size, int):
            short, long = (w, h) if w <= h else (h, w)
            new_short, new_long = self.size, int(self.size * long / short)
            nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        else:
            nh, nw = int(self.size[0]), int(self.size[1])
        if (nw, nh) == (w, h):
            return img
        return img.resize((nw, nh), Image.BILINEAR)


class ToTensor:
    """PIL RGB / uint8 HWC ndarray -> float32 CHW in [0, 1]."""

    def __call__(self, pic) -> torch.Tensor:
        arr = np.array(pic)     # a writable copy
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
        return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)


class RandomCrop:
    """transforms.RandomCrop(size=(h, w), padding, pad_if_needed) on a PIL image, zero fill."""

    def __init__(self, size: Tuple[int, int], pad_if_needed: bool = False, padding: int = None):
        self.size, self.pad_if_needed, self.padding = (int(size[0]), int(size[1])), pad_if_needed, padding

    @staticmethod
    def _pad(img: Image.Image, left: int, top: int, right: int, bottom: int) -> Image.Image:
        w, h = img.size
        out = Image.new(img.mode, (w + left + right, h + top + bottom), 0)
        out.paste(img, (left, top))
        return out

    def __call__(self, img: Image.Image) -> Image.Image:
        if self.padding is not None:
            p = int(self.padding)
            img = self._pad(img, p, p, p, p)
        th, tw = self.size
        w, h = img.size
        if self.pad_if_needed and w < tw:
            img = self._pad(img, tw - w, 0, tw - w, 0)
        if self.pad_if_needed and h < th:
            img = self._pad(img, 0, th - h, 0, th - h)
        w, h = img.size
        if h < th or w < tw:
            raise ValueError(f"Required crop size {(th, tw)} is larger than input image size {(h, w)}")
        i = int(torch.randint(0, h - th + 1, size=(1,)).item())
        j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
        return img.crop((j, i, j + tw, i + th))


class RandomRotation:
    """transforms.RandomRotation(degrees) on a CHW tensor: angle ~ U(-d, d), rotation about the centre,
    nearest interpolation, same canvas, zero fill."""

    def __init__(self, degrees: float):
        self.degrees = (-float(degrees), float(degrees))

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        return rotate(img, float(torch.empty(1).uniform_(self.degrees[0], self.degrees[1]).item()))


def rotate(img: torch.Tensor, angle: float) -> torch.Tensor:
    """Rotation of a CHW tensor by ``angle`` degrees about the centre: nearest, same canvas, zero fill."""
    c, h, w = img.shape
    a = math.radians(angle)
    # inverse map output -> input in pixel units about the centre, then to normalised coords
    cos, sin = math.cos(a), math.sin(a)
    theta = torch.tensor([[cos, sin * h / w, 0.0], [-sin * w / h, cos, 0.0]], dtype=img.dtype).unsqueeze(0)
    grid = F.affine_grid(theta, (1, c, h, w), align_corners=False)
    return F.grid_sample(img.unsqueeze(0), grid, mode="nearest", padding_mode="zeros", align_corners=False)[0]


def _gray(img: torch.Tensor) -> torch.Tensor:
    r, g, b = img.unbind(0)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(0)


def _blend(a: torch.Tensor, b: torch.Tensor, ratio: float) -> torch.Tensor:
    return (ratio * a + (1.0 - ratio) * b).clamp(0, 1)


def adjust_sharpness(img: torch.Tensor, factor: float) -> torch.Tensor:
    if img.shape[-1] <= 2 or img.shape[-2] <= 2:
        return img
    k = torch.ones(3, 3, dtype=img.dtype)
    k[1, 1] = 5.0
    k = (k / k.sum()).expand(img.shape[0], 1, 3, 3)
    blurred = img.clone()
    blurred[:, 1:-1, 1:-1] = F.conv2d(img.unsqueeze(0), k, groups=img.shape[0])[0]
    return _blend(img, blurred, factor)


class RandomAdjustSharpness:
    def __init__(self, sharpness_factor: float, p: float = 0.5):
        self.factor, self.p = sharpness_factor, p

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        return adjust_sharpness(img, self.factor) if torch.rand(1).item() < self.p else img


def _rgb2hsv(img):
    r, g, b = img.unbind(0)
    maxc, minc = img.max(0).values, img.min(0).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc))


def _hsv2rgb(img):
    h, s, v = img.unbind(0)
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p = (v * (1.0 - s)).clamp(0, 1)
    q = (v * (1.0 - s * f)).clamp(0, 1)
    t = (v * (1.0 - s * (1.0 - f))).clamp(0, 1)
    mask = i.unsqueeze(0) == torch.arange(6).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v))
    a2 = torch.stack((t, v, v, q, p, p))
    a3 = torch.stack((p, p, t, v, v, q))
    a4 = torch.stack((a1, a2, a3))
    return torch.einsum("ijk,xijk->xjk", mask.to(img.dtype), a4)


class ColorJitter:
    """transforms.ColorJitter(brightness, contrast, saturation, hue): the four adjustments in a random order."""

    def __init__(self, brightness=0.0, contrast=0.0, saturation=0.0, hue=0.0):
        self.brightness = (max(0.0, 1 - brightness), 1 + brightness) if brightness else None
        self.contrast = (max(0.0, 1 - contrast), 1 + contrast) if contrast else None
        self.saturation = (max(0.0, 1 - saturation), 1 + saturation) if saturation else None
        self.hue = (-hue, hue) if hue else None

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        order = torch.randperm(4)
        u = lambda rng: None if rng is None else float(torch.empty(1).uniform_(rng[0], rng[1]))
        b, c, s, h = u(self.brightness), u(self.contrast), u(self.saturation), u(self.hue)
        return color_jitter(img, order.tolist(), b, c, s, h)


def color_jitter(img: torch.Tensor, order, b=None, c=None, s=None, h=None) -> torch.Tensor:
    """ColorJitter with its draws made explicit: ``order`` = permutation of (0 brightness, 1 contrast, 2 saturation,
    3 hue), factors b / c / s and hue shift h (None = that adjustment is off)."""
    for fn in order:
        if fn == 0 and b is not None:
            img = _blend(img, torch.zeros_like(img), b)
        elif fn == 1 and c is not None:
            img = _blend(img, _gray(img).mean(), c)
        elif fn == 2 and s is not None:
            img = _blend(img, _gray(img).expand_as(img), s)
        elif fn == 3 and h is not None:
            hsv = _rgb2hsv(img)
            hsv[0] = (hsv[0] + h) % 1.0
            img = _hsv2rgb(hsv)
    return img


class RandomHorizontalFlip:
    def __init__(self, p: float = 0.5):
        self.p = p

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        return img.flip(-1) if torch.rand(1).item() < self.p else img


class AddGaussianNoise:
    """src/utils/custom_transform.py:9-29 (numpy RNG, min-max renormalisation, uint8 round trip)."""

    def __init__(self, mean=0., sigma_range=(0., .06), p=0.5):
        self.mean, self.sigma_range, self.p = mean, sigma_range, p

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        if np.random.rand() < self.p:
            c, h, w = img.squeeze().shape
            sigma = np.random.uniform(*self.sigma_range)
            return add_noise(img, np.random.normal(self.mean, sigma, (h, w, c)))
        return img


def add_noise(img: torch.Tensor, noise_hwc: np.ndarray) -> torch.Tensor:
    """AddGaussianNoise with its noise field made explicit (HWC): min-max to [0, 1], add, min-max to [0, 255],
    truncate to uint8, back to a [0, 1] tensor."""
    arr = img.squeeze().permute(1, 2, 0).detach().cpu().numpy()
    arr = np.interp(arr, (arr.min(), arr.max()), (0, 1))
    noisy = arr + noise_hwc
    noisy = np.interp(noisy, (noisy.min(), noisy.max()), (0, 255)).astype(np.uint8)
    return ToTensor()(noisy)


class RGBToGrayscaleStacked:
    """src/utils/custom_transform.py:32-46 (weights 0.299 / 0.587 / 0.144 as written there)."""

    def __init__(self, enable: bool = False):
        self.enable = enable

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        if self.enable:
            g = img[0:1] * 0.299 + img[1:2] * 0.587 + img[2:3] * 0.144
            return torch.cat([g] * 3, dim=0)
        return img
