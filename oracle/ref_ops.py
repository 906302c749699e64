"""CPU oracle: plain-torch fp32 restatement of the reference's AdaIN hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``style_transfer_amd/`` imports this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the CPU baseline.

Every function cites the reference ``file:line`` (relative to the upstream
checkout) it restates.  The restatement is pinned against the reference itself:
``oracle/gen_golden.py`` imports the reference's own modules in the build
container, runs them on seeded inputs and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against those vectors.

Parity status: pinned by reference-generated golden vectors for generator.py,
transformer.py and losses.py.  The loss network (torchvision VGG19 +
weight_norm + IntermediateLayerGetter, trainer.py:165-199) cannot be imported
here (torchvision absent) -> that part is "parity unpinned" and follows the
documented torchvision 0.15.1 behaviour (see ``loss_vgg_features``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# Layer tables
# --------------------------------------------------------------------------

#: Encoder Sequential (generator.py:45-77): state-dict index -> (cin, cout, k)
ENCODER_CONVS: List[Tuple[int, int, int, int]] = [
    (0, 3, 3, 1), (2, 3, 64, 3), (5, 64, 64, 3), (9, 64, 128, 3), (12, 128, 128, 3),
    (16, 128, 256, 3), (19, 256, 256, 3), (22, 256, 256, 3), (25, 256, 256, 3),
    (29, 256, 512, 3),
]
#: ceil-mode 2x2 max-pools follow these encoder conv indices (generator.py:53,60,73)
ENCODER_POOL_AFTER = (5, 12, 25)

#: Decoder Sequential (generator.py:88-118): index -> (cin, cout); nearest x2
#: upsample follows indices 1, 14, 21; the last conv (28) has no ReLU.
DECODER_CONVS: List[Tuple[int, int, int]] = [
    (1, 512, 256), (5, 256, 256), (8, 256, 256), (11, 256, 256), (14, 256, 128),
    (18, 128, 128), (21, 128, 64), (25, 64, 64), (28, 64, 3),
]
DECODER_UP_AFTER = (1, 14, 21)

#: torchvision vgg19 cfg "E" `.features` (trainer.py:168): index -> op
VGG19_FEATURES: List[Tuple[str, int, int]] = []


def _build_vgg19_table() -> None:
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
           512, 512, 512, 512, "M"]
    cin = 3
    for v in cfg:
        if v == "M":
            VGG19_FEATURES.append(("pool", 0, 0))
        else:
            VGG19_FEATURES.append(("conv", cin, int(v)))
            VGG19_FEATURES.append(("relu", 0, 0))
            cin = int(v)


_build_vgg19_table()

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


# --------------------------------------------------------------------------
# conv operand precision (fp32 = the reference; "bf16" emulates the HIP bf16 compute mode)
# --------------------------------------------------------------------------

CONV_COMPUTE = "f32"


class conv_compute:
    """Context manager: ``with conv_compute("bf16"):`` makes every 3x3 convolution of this module
    round its operands (activations, weights, and the output gradient in the data-gradient) to
    bf16 with fp32 accumulation -- the numerics of style_transfer_amd's bf16 compute mode
    (csrc/conv3x3_bf16.hip, conv3x3_wgrad.hip); the bias gradient sums the unrounded gradient."""

    def __init__(self, name: str):
        assert name in ("f32", "bf16")
        self.name = name

    def __enter__(self):
        global CONV_COMPUTE
        self.prev, CONV_COMPUTE = CONV_COMPUTE, self.name

    def __exit__(self, *exc):
        global CONV_COMPUTE
        CONV_COMPUTE = self.prev


def _round_bf16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(t.dtype)      # dtype-preserving: the float64 arbiter runs of the tests keep float64


class _ConvBf16Operands(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, w, b):
        ctx.save_for_backward(xp, w)
        return F.conv2d(_round_bf16(xp), _round_bf16(w), b)

    @staticmethod
    def backward(ctx, gy):
        xp, w = ctx.saved_tensors
        gx = torch.nn.grad.conv2d_input(xp.shape, _round_bf16(w), _round_bf16(gy)) if ctx.needs_input_grad[0] else None
        gw = torch.nn.grad.conv2d_weight(_round_bf16(xp), w.shape, _round_bf16(gy)) if ctx.needs_input_grad[1] else None
        gb = gy.sum((0, 2, 3)) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def conv3x3_padded(xp: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d(cin, cout, (3,3)) on an already padded input (generator.py:48 etc.)."""
    if CONV_COMPUTE == "bf16":
        return _ConvBf16Operands.apply(xp, w, b)
    return F.conv2d(xp, w, b)


# --------------------------------------------------------------------------
# gates: the piecewise-linear switches of the training step (ReLU masks, max-pool winners, the [0, 1] clamp of
# losses.py:170).  Two fp32 implementations of the step agree on all but a few isolated switches (a 1e-7 forward
# difference flips one now and then) and everything below a flipped switch moves, so whole-step GRADIENTS are
# compared with the switches pinned: ``with gates("record") as g:`` stores the switches this oracle takes under
# their names, ``with gates("force", g):`` makes it take the stored ones (from its own run, or from the other
# implementation's activations) instead of deciding itself.  Forced forward values are the linear branch's
# (``x * mask``, the chosen window element, ``x`` inside the clamp / the constant outside); gradients follow the
# mask exactly.  Names: "dec.{idx}" (decoder ReLUs, generator.py:89-116), "{tag}.relu.{idx}" / "{tag}.pool.{idx}"
# (loss VGG, trainer.py:165-199), "hist" (losses.py:170).  Test infrastructure; the reference has no such thing.
# --------------------------------------------------------------------------

GATES = None        # None | ("record", dict) | ("force", dict)


class gates:
    def __init__(self, mode: str, store: dict = None):
        assert mode in ("record", "force")
        self.mode, self.store = mode, ({} if store is None else store)

    def __enter__(self):
        global GATES
        self.prev, GATES = GATES, (self.mode, self.store)
        return self.store

    def __exit__(self, *exc):
        global GATES
        GATES = self.prev


def _relu(x: torch.Tensor, name=None) -> torch.Tensor:
    if GATES is None or name is None:
        return F.relu(x)
    mode, store = GATES
    if mode == "record":
        store[name] = (x > 0).detach()
        return F.relu(x)
    return x * store[name].to(x.dtype)


def _max_pool(x: torch.Tensor, name=None, ceil_mode: bool = False) -> torch.Tensor:
    if GATES is None or name is None:
        return F.max_pool2d(x, 2, 2, 0, ceil_mode=ceil_mode)
    mode, store = GATES
    if mode == "record":
        y, idx = F.max_pool2d(x, 2, 2, 0, ceil_mode=ceil_mode, return_indices=True)
        store[name] = idx.detach()
        return y
    idx = store[name]
    return x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)


def _clamp01(x: torch.Tensor, name=None) -> torch.Tensor:
    if GATES is None or name is None:
        return torch.clamp(x, 0, 1)
    mode, store = GATES
    if mode == "record":
        store[name] = ((x >= 0) & (x <= 1)).detach(), (x > 1).detach()      # torch.clamp passes the gradient on [0, 1]
        return torch.clamp(x, 0, 1)
    inside, above = store[name]
    return torch.where(inside, x, above.to(x.dtype))


# --------------------------------------------------------------------------
# generator.py
# --------------------------------------------------------------------------

def symmetric_pad_index(n: int, lo: int, hi: int) -> np.ndarray:
    """Index map of SymmetricPadding2D along one axis (generator.py:23-37).

    Triangular-wave reflection of ``arange(-lo, n+hi)`` about the half-pixel
    points -0.5 and n-0.5 (numpy "symmetric": the edge pixel is repeated).
    Integer restatement of the float fmod arithmetic: period 2n,
    ``m = (i mod 2n)``, index = m if m < n else 2n-1-m.
    """
    i = np.arange(-lo, n + hi)
    m = np.mod(i, 2 * n)
    return np.where(m >= n, 2 * n - 1 - m, m).astype(np.int64)


def symmetric_pad(x: torch.Tensor, padding=(1, 1, 1, 1)) -> torch.Tensor:
    """SymmetricPadding2D.forward (generator.py:19-39). padding = (l, r, t, b)."""
    h, w = x.shape[-2:]
    left, right, top, bottom = padding
    xi = torch.from_numpy(symmetric_pad_index(w, left, right))
    yi = torch.from_numpy(symmetric_pad_index(h, top, bottom))
    return x[..., yi[:, None], xi[None, :]]


def encoder_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Encoder.forward (generator.py:42-82).  ``sd`` uses the reference's bare
    Sequential keys ``"0.weight" ... "29.bias"``."""
    for idx, _cin, _cout, k in ENCODER_CONVS:
        w, b = sd[f"{idx}.weight"], sd[f"{idx}.bias"]
        if k == 1:
            x = F.conv2d(x, w, b)
        else:
            x = F.relu(conv3x3_padded(symmetric_pad(x), w, b))
        if idx in ENCODER_POOL_AFTER:
            x = F.max_pool2d(x, 2, 2, 0, ceil_mode=True)
    return x


def decoder_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str = "decoder.",
                    gate_tag: str = None) -> torch.Tensor:
    """Decoder.forward (generator.py:85-130).  Keys ``decoder.{1,...,28}.{weight,bias}``.  ``gate_tag``: see gates."""
    for idx, _cin, _cout in DECODER_CONVS:
        w, b = sd[f"{prefix}{idx}.weight"], sd[f"{prefix}{idx}.bias"]
        x = conv3x3_padded(symmetric_pad(x), w, b)
        if idx != 28:
            x = _relu(x, f"{gate_tag}.{idx}" if gate_tag else None)
        if idx in DECODER_UP_AFTER:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
    return x


def make_encoder_state(seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded synthetic stand-in for the missing vgg_normalised_conv4_1.pth
    (He-scaled normals; same keys/shapes as generator.py:46-75)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for idx, cin, cout, k in ENCODER_CONVS:
        std = math.sqrt(2.0 / (cin * k * k))
        sd[f"{idx}.weight"] = (torch.randn(cout, cin, k, k, generator=g) * std).to(dtype)
        sd[f"{idx}.bias"] = (torch.randn(cout, generator=g) * 0.05).to(dtype)
    return sd


def make_decoder_state(seed: int = 1, prefix: str = "decoder.", dtype=torch.float32) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for idx, cin, cout in DECODER_CONVS:
        std = math.sqrt(2.0 / (cin * 9))
        sd[f"{prefix}{idx}.weight"] = (torch.randn(cout, cin, 3, 3, generator=g) * std).to(dtype)
        sd[f"{prefix}{idx}.bias"] = (torch.randn(cout, generator=g) * 0.05).to(dtype)
    return sd


# --------------------------------------------------------------------------
# transformer.py
# --------------------------------------------------------------------------

def calc_mean_std(feat: torch.Tensor, eps=1e-5):
    """AdaIN.calc_mean_std (transformer.py:15-24): unbiased var + eps, sqrt; mean."""
    assert feat.dim() == 4
    n, c = feat.shape[:2]
    var = feat.reshape(n, c, -1).var(dim=2) + eps
    std = var.sqrt().view(n, c, 1, 1)
    mean = feat.reshape(n, c, -1).mean(dim=2).view(n, c, 1, 1)
    return mean, std


def adjust_sigmoid(p: torch.Tensor, upper_limit: float = 0.3):
    """AdaIN.adjust_sigmoid (transformer.py:27-36): sqrt(|p|) clamped to
    [1e-5, upper_limit]; the clamped branches return Python floats (no grad)."""
    lower_limit = 1e-5
    x = torch.pow(torch.abs(p), 0.5)
    if x < lower_limit:
        return lower_limit
    if x > upper_limit:
        return upper_limit
    return x


def adain(content: torch.Tensor, style: torch.Tensor, style_eps=1e-5, content_eps=1e-5) -> torch.Tensor:
    """AdaIN.adaptive_instance_normalization (transformer.py:46-54)."""
    assert content.shape[:2] == style.shape[:2]
    s_mean, s_std = calc_mean_std(style, style_eps)
    c_mean, c_std = calc_mean_std(content, content_eps)
    return (content - c_mean) / c_std * s_std + s_mean


def adain_forward(content, style, content_eps_param, style_eps_param, eps_limit=1e-5):
    """AdaIN.forward (transformer.py:56-59)."""
    return adain(content, style,
                 style_eps=adjust_sigmoid(style_eps_param, eps_limit),
                 content_eps=adjust_sigmoid(content_eps_param, eps_limit))


def blend(t: torch.Tensor, fc: torch.Tensor, alpha: float) -> torch.Tensor:
    """trainer.py:434 / :702."""
    return alpha * t + (1 - alpha) * fc


# --------------------------------------------------------------------------
# trainer.py pieces (restated; trainer.py itself needs torchvision + wandb)
# --------------------------------------------------------------------------

def normalize(x: torch.Tensor, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
    """transforms.Normalize (trainer.py:299, 440-442, 707-708): (x - m) / s per channel."""
    m = torch.tensor(mean, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
    s = torch.tensor(std, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
    return (x - m) / s


def weight_norm_effective(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """torch.nn.utils.weight_norm(dim=0) (trainer.py:179-181): w = g * v / ||v||,
    norm over every dim but 0.  g is initialised to ||v||."""
    norm = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1, 1)
    return v * (g.view(-1, 1, 1, 1) / norm)


def make_loss_vgg_state(seed: int = 2, upto: int = 36, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded synthetic torchvision-vgg19 ``features`` state dict ("{idx}.weight")."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for idx, (kind, cin, cout) in enumerate(VGG19_FEATURES):
        if kind == "conv" and idx <= upto:
            std = math.sqrt(2.0 / (cin * 9))
            sd[f"{idx}.weight"] = (torch.randn(cout, cin, 3, 3, generator=g) * std).to(dtype)
            sd[f"{idx}.bias"] = (torch.randn(cout, generator=g) * 0.05).to(dtype)
    return sd


def loss_vgg_features(x: torch.Tensor, sd: Dict[str, torch.Tensor], indices: Sequence[int],
                      gate_tag: str = None) -> List[torch.Tensor]:
    """Trainer.get_feature_extractor + IntermediateLayerGetter (trainer.py:165-199,
    446-454).  PARITY UNPINNED (torchvision not importable here).

    torchvision vgg19.features: zero-pad 3x3 convs, ReLU(inplace=True), floor
    2x2 max-pools.  The getter truncates after the last requested index and
    returns the tensors produced at the requested indices, in layer order.
    Because the ReLUs are in-place, a captured *conv* output is overwritten by
    the ReLU that follows it, unless the conv is the last layer kept.  Weights
    are used as given (weight_norm with g=||v|| is the identity up to rounding;
    see ``weight_norm_effective``).
    """
    wanted = sorted(set(int(i) for i in indices))
    last = max(wanted)
    outs: Dict[int, torch.Tensor] = {}
    for idx, (kind, _cin, _cout) in enumerate(VGG19_FEATURES[: last + 1]):
        if kind == "conv":
            x = conv3x3_padded(F.pad(x, (1, 1, 1, 1)), sd[f"{idx}.weight"], sd[f"{idx}.bias"])
        elif kind == "relu":
            x = _relu(x, f"{gate_tag}.relu.{idx}" if gate_tag else None)
            if idx - 1 in outs:          # in-place aliasing of the captured conv output
                outs[idx - 1] = x
        else:
            x = _max_pool(x, f"{gate_tag}.pool.{idx}" if gate_tag else None)
        if idx in wanted:
            outs[idx] = x
    # dict order of IntermediateLayerGetter = layer order; trainer indexes the
    # resulting list by position in the user's index list (trainer.py:448-457),
    # which matches only for ascending lists -- keep the reference's behaviour.
    return [outs[i] for i in wanted]


# --------------------------------------------------------------------------
# losses.py
# --------------------------------------------------------------------------

def adain_style_loss(inp: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """AdaINStyleLoss (losses.py:14-39): MSE(mean) + MSE(std), eps = 1e-5."""
    assert inp.shape == tgt.shape
    im, is_ = calc_mean_std(inp)
    tm, ts = calc_mean_std(tgt)
    return F.mse_loss(im, tm) + F.mse_loss(is_, ts)


def calc_2_moments(x: torch.Tensor):
    """VincentStyleLoss.calc_2_moments (losses.py:49-54): mean and UN-normalised covariance."""
    b, c, w, h = x.shape
    x = x.reshape(b, c, w * h)
    mu = x.mean(dim=-1, keepdim=True)
    cov = torch.bmm(x - mu, torch.transpose(x - mu, -1, -2))
    return mu, cov


def l2wass_dist(mean_stl, cov_stl, mean_synth, cov_synth):
    """VincentStyleLoss.l2wass_dist (losses.py:67-81)."""
    eigvals, eigvects = torch.linalg.eigh(cov_stl)
    root = torch.diag_embed(torch.sqrt(eigvals.clip(0)))
    root_cov_stl = eigvects @ root @ eigvects.transpose(-1, -2)
    tr_cov_stl = torch.sum(eigvals.clip(0), dim=1, keepdim=True)
    tr_cov_synth = torch.sum(torch.linalg.eigvalsh(cov_synth).clip(0), dim=1, keepdim=True)
    mean_diff_squared = torch.mean((mean_synth - mean_stl) ** 2)
    cov_prod = root_cov_stl @ cov_synth @ root_cov_stl
    var_overlap = torch.sum(torch.sqrt(torch.linalg.eigvalsh(cov_prod).clip(0.1)), dim=1, keepdim=True)
    return mean_diff_squared + tr_cov_stl + tr_cov_synth - 2 * var_overlap


def vincent_style_loss(inp: torch.Tensor, tgt: torch.Tensor, scale_factor: float = 1e-5) -> torch.Tensor:
    """VincentStyleLoss.forward (losses.py:83-90)."""
    mean_synth, cov_synth = calc_2_moments(inp)
    mean_stl, cov_stl = calc_2_moments(tgt)
    return l2wass_dist(mean_stl, cov_stl, mean_synth, cov_synth).mean() * scale_factor


def content_loss(inp: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """AdaINContentLoss (losses.py:93-101)."""
    assert inp.shape == tgt.shape
    return F.mse_loss(inp, tgt)


def tv_loss(img: torch.Tensor) -> torch.Tensor:
    """TVLoss.forward (losses.py:104-116)."""
    b, c, h, w = img.shape
    count_h = c * (h - 1) * w
    count_w = c * h * (w - 1)
    h_tv = ((img[:, :, 1:, :] - img[:, :, : h - 1, :]) ** 2).div(count_h)
    w_tv = ((img[:, :, :, 1:] - img[:, :, :, : w - 1]) ** 2).div(count_w)
    return (2 * (h_tv.sum() + w_tv.sum()) / b) * 100


HIST_EPS = 1e-6  # losses.py:119
#: dtype the kernel weights are cast to before the [h,N]x[N,h] product: float32 in the reference
#: (losses.py:239-240).  Tests that use a float64 run of this oracle as an ARBITER between two fp32
#: implementations switch it to float64 (``with hist_kernel_dtype(torch.float64)``) so that the whole
#: computation is double precision; nothing else changes.
HIST_KERNEL_DTYPE = torch.float32


class hist_kernel_dtype:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global HIST_KERNEL_DTYPE
        self.prev, HIST_KERNEL_DTYPE = HIST_KERNEL_DTYPE, self.dtype

    def __exit__(self, *exc):
        global HIST_KERNEL_DTYPE
        HIST_KERNEL_DTYPE = self.prev


def rgbuv_hist(x: torch.Tensor, h: int = 64, insz: int = 128, method: str = "inverse-quadratic",
               sigma: float = 0.02, intensity_scale: bool = True,
               hist_boundary=(-3.0, 3.0), resizing: str = "interpolation", green_only: bool = False,
               gate_name: str = None) -> torch.Tensor:
    """RGBuvHistBlock.forward (losses.py:169-322).  Bin centres are float64 (np.linspace, losses.py:211-213),
    so the kernel weights are evaluated in float64 and cast to float32
    (losses.py:239-240) before the [h,N]x[N,h] product.  'sampling' picks h (sic: the bin count, losses.py:176-181)
    rows and columns; green_only keeps the G pairing alone (losses.py:195-196, 278-281)."""
    lo, hi = sorted(hist_boundary)
    x = _clamp01(x, gate_name)
    if x.shape[2] > insz or x.shape[3] > insz:
        if resizing == "interpolation":
            x = F.interpolate(x, size=(insz, insz), mode="bilinear", align_corners=False)
        elif resizing == "sampling":
            i1 = torch.LongTensor(np.linspace(0, x.shape[2], h, endpoint=False))
            i2 = torch.LongTensor(np.linspace(0, x.shape[3], h, endpoint=False))
            x = x.index_select(2, i1).index_select(3, i2)
        else:
            raise Exception(f"Wrong resizing method {resizing}")
    if x.shape[1] > 3:
        x = x[:, :3]
    bins = torch.tensor(np.linspace(lo, hi, num=h)).unsqueeze(0)       # float64 [1,h]
    thr_eps = (abs(lo) + abs(hi)) / h
    hists = []
    for l in range(x.shape[0]):
        I = x[l].reshape(3, -1).t()                                     # [N,3]
        if intensity_scale:
            Iy = torch.sqrt((I ** 2).sum(dim=1) + HIST_EPS).unsqueeze(1)
        else:
            Iy = 1
        logI = torch.log(I + HIST_EPS)
        per_ch = []
        for ch in ((1,) if green_only else range(3)):
            others = [o for o in range(3) if o != ch]
            Iu = (logI[:, ch] - logI[:, others[0]]).unsqueeze(1)
            Iv = (logI[:, ch] - logI[:, others[1]]).unsqueeze(1)
            du = (Iu - bins).abs()
            dv = (Iv - bins).abs()
            if method == "thresholding":
                du = du <= thr_eps / 2
                dv = dv <= thr_eps / 2
            elif method == "RBF":
                du = torch.exp(-(du ** 2) / sigma ** 2)
                dv = torch.exp(-(dv ** 2) / sigma ** 2)
            elif method == "inverse-quadratic":
                du = 1 / (1 + du ** 2 / sigma ** 2)
                dv = 1 / (1 + dv ** 2 / sigma ** 2)
            else:
                raise Exception(f"Wrong kernel method {method}")
            du = du.type(HIST_KERNEL_DTYPE)
            dv = dv.type(HIST_KERNEL_DTYPE)
            per_ch.append(torch.mm((Iy * du).t(), dv))
        hists.append(torch.stack(per_ch))
    hists = torch.stack(hists)
    return hists / (hists.sum(dim=(1, 2, 3)).view(-1, 1, 1, 1) + HIST_EPS)


def hist_loss(inp: torch.Tensor, tgt: torch.Tensor, gate_name: str = None, **kw) -> torch.Tensor:
    """HistLoss.forward (losses.py:338-347): Hellinger, sqrt OUTSIDE the batch sum."""
    ih = rgbuv_hist(inp, gate_name=gate_name, **kw)
    th = rgbuv_hist(tgt, **kw)
    return (1 / np.sqrt(2.0) * torch.sqrt(torch.sum((torch.sqrt(th) - torch.sqrt(ih)) ** 2)) / ih.shape[0]) * 100


def compute_loss(content_feats, style_feats, stylized, style_imgs, content_imgs,
                 alpha: float, beta: float, gamma: float, delta: float, use_vincent: bool = False):
    """Trainer.compute_loss (trainer.py:246-279). *_feats: list of (model_output, original)."""
    style_fn = vincent_style_loss if use_vincent else adain_style_loss
    loss_content = sum(content_loss(o, t) for o, t in content_feats)
    loss_style = sum(style_fn(o, t) for o, t in style_feats)
    variation = tv_loss(stylized)
    hist = 0.4 * hist_loss(stylized, style_imgs, gate_name="hist") + 0.6 * hist_loss(stylized, content_imgs, gate_name="hist")
    total = alpha * loss_content + beta * loss_style + gamma * variation + delta * hist
    return loss_content, loss_style, variation, hist, total


def stylize(content: torch.Tensor, style: torch.Tensor, enc_sd, dec_sd,
            content_eps_param=None, style_eps_param=None, eps_limit: float = 1e-5,
            alpha: float = 1.0, do_normalize: bool = False) -> torch.Tensor:
    """Inference composition, Trainer.plot_comparison (trainer.py:698-708)."""
    if content_eps_param is None:
        content_eps_param = torch.tensor([1e-5])
    if style_eps_param is None:
        style_eps_param = torch.tensor([1e-5])
    fc = encoder_forward(content, enc_sd)
    fs = encoder_forward(style, enc_sd)
    t = adain_forward(fc, fs, content_eps_param, style_eps_param, eps_limit)
    t = blend(t, fc, alpha)
    out = decoder_forward(t, dec_sd)
    return normalize(out) if do_normalize else out


def train_step_losses(content, style, enc_sd, dec_sd, vgg_sd, content_eps_param, style_eps_param,
                      eps_limit, content_layers_idx, style_layers_idx,
                      alpha, beta, gamma, delta, use_vincent=False, decoded_override=None):
    """Forward half of the train-step body (trainer.py:426-464).

    ``decoded_override`` (tests only): the decoded image of ANOTHER implementation of the same step.  Everything
    after the decoder is evaluated at those values while gradients still flow into this decoder
    (``dec + (override - dec).detach()``): the loss terms whose gradient is very sensitive to the last bits of the
    decoded image (the histogram's 1 / (x + 1e-6)) can then be compared at identical input."""
    fc = encoder_forward(content, enc_sd)
    fs = encoder_forward(style, enc_sd)
    t = adain_forward(fc, fs, content_eps_param, style_eps_param, eps_limit)
    t = 1.0 * t + (1 - 1.0) * fc
    dec = decoder_forward(t, dec_sd, gate_tag="dec")
    if decoded_override is not None:
        dec = dec + (decoded_override.to(dec.dtype) - dec).detach()
    c_n, s_n, d_n = normalize(content), normalize(style), normalize(dec)
    # (gates: only the passes a gradient flows through carry names -- the decoder and the decoded image's two
    # extractor passes, which see the same input and therefore the same switches)
    cf = list(zip(loss_vgg_features(d_n, vgg_sd, content_layers_idx, gate_tag="vgg"),
                  loss_vgg_features(c_n, vgg_sd, content_layers_idx)))
    sf = list(zip(loss_vgg_features(d_n, vgg_sd, style_layers_idx, gate_tag="vgg"),
                  loss_vgg_features(s_n, vgg_sd, style_layers_idx)))
    return compute_loss(cf, sf, d_n, s_n, c_n, alpha, beta, gamma, delta, use_vincent)
