"""Loss modules with the reference's names and constructor defaults (src/models/losses.py),
forward AND backward on HIP kernels.  Gradients flow to the first argument (``input`` /
``image``); the second (``target``) is treated as a constant, which is all the reference's
training step needs (targets are features of the un-trained content/style images,
trainer.py:446-457)."""
from __future__ import annotations

import os
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops


def _workspace(dev):
    return torch.empty(_lib.load().st_reduce_workspace_bytes() // 4, dtype=torch.float32, device=dev)


def _gout_ptr(g):
    return _lib.dev_ptr(g.contiguous().reshape(1), "gout")


# --------------------------------------------------------------------------------------------
class _MSEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        _lib.check(lib.st_mse_fwd(_lib.dev_ptr(a, "input"), _lib.dev_ptr(b, "target"), a.numel(),
                                  _lib.dev_ptr(_workspace(a.device), "ws"), _lib.dev_ptr(out, "out"),
                                  _lib.stream_ptr()), "st_mse_fwd")
        ctx.save_for_backward(a, b)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        lib = _lib.load()
        da = torch.empty_like(a)
        g = g.contiguous().reshape(1)
        _lib.check(lib.st_mse_bwd(_lib.dev_ptr(a, "input"), _lib.dev_ptr(b, "target"), _lib.dev_ptr(g, "gout"), 1.0,
                                  _lib.dev_ptr(da, "da"), a.numel(), 0, _lib.stream_ptr()), "st_mse_bwd")
        return da, None


def mse(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    assert a.size() == b.size()
    return _MSEFunction.apply(a.contiguous(), b.detach().contiguous())


class AdaINContentLoss(nn.Module):
    """losses.py:93-101."""

    def forward(self, input, target):
        assert (input.size() == target.size())
        return mse(input, target)


# --------------------------------------------------------------------------------------------
class _MeanStdLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target):
        lib = _lib.load()
        ma, sa = ops.mean_std(x, 1e-5)
        mb, sb = ops.mean_std(target, 1e-5)
        planes = int(x.shape[0] * x.shape[1])
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        _lib.check(lib.st_meanstd_loss_fwd(_lib.dev_ptr(ma, "ma"), _lib.dev_ptr(sa, "sa"), _lib.dev_ptr(mb, "mb"),
                                           _lib.dev_ptr(sb, "sb"), planes, _lib.dev_ptr(_workspace(x.device), "ws"),
                                           _lib.dev_ptr(out, "out"), _lib.stream_ptr()), "st_meanstd_loss_fwd")
        ctx.save_for_backward(x, ma, sa, mb, sb)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        x, ma, sa, mb, sb = ctx.saved_tensors
        lib = _lib.load()
        dx = torch.empty_like(x)
        planes = int(x.shape[0] * x.shape[1])
        hw = int(x.shape[2] * x.shape[3])
        g = g.contiguous().reshape(1)
        _lib.check(lib.st_meanstd_loss_bwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(ma, "ma"), _lib.dev_ptr(sa, "sa"),
                                           _lib.dev_ptr(mb, "mb"), _lib.dev_ptr(sb, "sb"), _lib.dev_ptr(g, "gout"), 1.0,
                                           _lib.dev_ptr(dx, "dx"), planes, hw, 0, _lib.stream_ptr()),
                   "st_meanstd_loss_bwd")
        return dx, None


class _MeanStdLossTapFunction(torch.autograd.Function):
    """_MeanStdLossFunction for an input that is a loss TAP of the bf16 VGG chain (autograd_vgg.VGGChainFunction): the
    gradient w.r.t. the tap is k0 + k1 * (x - mean) per (image, channel) plane, and the chain's join kernel reads the
    tap anyway -- so the backward hands over the per-plane coefficients (autograd_vgg.lazy_affine_grad) instead of
    writing an [n, c, h, w] fp32 tensor that the join would read back: 12 bytes per tap element less HBM traffic
    (0.6 ms of a 15 ms bf16 step at batch 8, 512 x 512).  Same fp32 expression: identical values."""

    @staticmethod
    def forward(ctx, x, target):
        out = _MeanStdLossFunction.forward(ctx, x, target)
        return out

    @staticmethod
    def backward(ctx, g):
        from .autograd_vgg import lazy_affine_grad
        x, ma, sa, mb, sb = ctx.saved_tensors
        coeffs = ops.meanstd_loss_coeffs(ma, sa, mb, sb, g, int(x.shape[2] * x.shape[3]))
        return lazy_affine_grad(x, coeffs), None


class AdaINStyleLoss(nn.Module):
    """losses.py:14-39: MSE of channel means + MSE of channel stds (unbiased var + 1e-5)."""

    @staticmethod
    def calc_mean_std(feat, eps=1e-5):
        size = feat.size()
        assert (len(size) == 4)
        return ops.mean_std(feat, float(eps))

    def calc_style_loss(self, input, target):
        assert (input.size() == target.size())
        if getattr(input, "_st_chain_tap", False) and input.is_contiguous():
            return _MeanStdLossTapFunction.apply(input, target.detach().contiguous())
        return _MeanStdLossFunction.apply(input.contiguous(), target.detach().contiguous())

    def forward(self, input, target):
        return self.calc_style_loss(input, target)


# --------------------------------------------------------------------------------------------
class _TVFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        n, c, h, w = (int(v) for v in x.shape)
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        _lib.check(lib.st_tv_fwd(_lib.dev_ptr(x, "image"), n, c, h, w, _lib.dev_ptr(_workspace(x.device), "ws"),
                                 _lib.dev_ptr(out, "out"), _lib.stream_ptr()), "st_tv_fwd")
        ctx.save_for_backward(x)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        lib = _lib.load()
        n, c, h, w = (int(v) for v in x.shape)
        dx = torch.empty_like(x)
        g = g.contiguous().reshape(1)
        _lib.check(lib.st_tv_bwd(_lib.dev_ptr(x, "image"), _lib.dev_ptr(g, "gout"), 1.0, _lib.dev_ptr(dx, "dx"),
                                 n, c, h, w, 0, _lib.stream_ptr()), "st_tv_bwd")
        return dx


class TVLoss(nn.Module):
    """losses.py:104-116."""

    def forward(self, image):
        return _TVFunction.apply(image.contiguous())

    @staticmethod
    def _tensor_size(t):
        return t.size()[1] * t.size()[2] * t.size()[3]


# --------------------------------------------------------------------------------------------
EPS = 1e-6   # losses.py:119
_HIST_METHODS = {"inverse-quadratic": 0, "RBF": 1, "thresholding": 2}


class _RGBuvHistFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, insz, method, sigma, bins, intensity_scale):
        import ctypes
        lib = _lib.load()
        n, c, h, w = (int(v) for v in x.shape)
        hist = torch.empty((n, 3, 64, 64), dtype=torch.float32, device=x.device)
        totals = torch.empty(n, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.st_rgbuv_hist_workspace_bytes(n, h, w, insz) // 4, dtype=torch.float32, device=x.device)
        barr = (ctypes.c_double * len(bins))(*[float(b) for b in bins])
        _lib.check(lib.st_rgbuv_hist_fwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(hist, "hist"), _lib.dev_ptr(totals, "totals"),
                                         _lib.dev_ptr(ws, "workspace"), n, c, h, w, insz, method, float(sigma),
                                         ctypes.cast(barr, ctypes.c_void_p), len(bins), int(intensity_scale),
                                         _lib.stream_ptr()), "st_rgbuv_hist_fwd")
        ctx.cfg = (insz, method, sigma, tuple(bins), intensity_scale)
        ctx.save_for_backward(x, hist, totals)
        ctx.mark_non_differentiable(totals)
        return hist, totals                 # totals: un-normalised per-image sums (a constant for autograd)

    @staticmethod
    def backward(ctx, ghist, _gtotals=None):
        import ctypes
        x, hist, totals = ctx.saved_tensors
        insz, method, sigma, bins, intensity_scale = ctx.cfg
        lib = _lib.load()
        n, c, h, w = (int(v) for v in x.shape)
        dx = torch.empty_like(x)
        ws = torch.empty(lib.st_rgbuv_hist_workspace_bytes(n, h, w, insz) // 4, dtype=torch.float32, device=x.device)
        barr = (ctypes.c_double * len(bins))(*bins)
        ghist = ghist.contiguous()
        _lib.check(lib.st_rgbuv_hist_bwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(hist, "hist"), _lib.dev_ptr(totals, "totals"),
                                         _lib.dev_ptr(ghist, "ghist"), None, 1.0, _lib.dev_ptr(dx, "dx"),
                                         _lib.dev_ptr(ws, "workspace"), n, c, h, w, insz, method, float(sigma),
                                         ctypes.cast(barr, ctypes.c_void_p), len(bins), int(intensity_scale), 0,
                                         _lib.stream_ptr()), "st_rgbuv_hist_bwd")
        return dx, None, None, None, None, None


class RGBuvHistBlock(nn.Module):
    """losses.py:122-322 on the HIP histogram kernels (csrc/hist.hip).  HistLoss builds h=64, 'interpolation', all
    three pairings (losses.py:333-336); the other settings of the reference's block are served around the same
    kernels: h < 64 (the kernels' 64 x 64 bins with the unused centres parked where every weight is 0, sliced
    here), green_only (the G pairing's plane, re-normalised by its own sum, losses.py:195-196, 278-281, 319-320)
    and resizing='sampling' (h -- sic -- rows and columns picked by index, losses.py:175-183).  h > 64 raises."""

    def __init__(self, h=64, insz=150, resizing='interpolation', method='inverse-quadratic', sigma=0.02,
                 intensity_scale=True, hist_boundary=None, green_only=False, device='cuda'):
        super().__init__()
        self.h = h
        self.insz = insz
        self.device = device
        self.resizing = resizing
        self.method = method
        self.intensity_scale = intensity_scale
        self.green_only = green_only
        if hist_boundary is None:
            hist_boundary = [-3, 3]
        hist_boundary.sort()
        self.hist_boundary = hist_boundary
        if self.method == 'thresholding':
            self.eps = (abs(hist_boundary[0]) + abs(hist_boundary[1])) / h
        self.sigma = sigma
        if not 1 <= h <= 64:
            raise NotImplementedError("HIP RGBuvHistBlock supports 1 <= h <= 64 bins")
        if method not in _HIST_METHODS:
            raise Exception(f'Wrong kernel method. It should be either thresholding, RBF,'
                            f' inverse-quadratic. But the given value is {self.method}.')
        # float64 bin centres exactly as the reference builds them (losses.py:211-213)
        self._bins = [float(v) for v in np.linspace(self.hist_boundary[0], self.hist_boundary[1], num=self.h)]

    def forward(self, x):
        insz = int(self.insz)
        if x.shape[2] > insz or x.shape[3] > insz:
            if self.resizing == 'sampling':
                # clamp (inside the kernel) and index selection commute; the picked image is never resized again
                i1 = torch.as_tensor(np.linspace(0, x.shape[2], self.h, endpoint=False).astype(np.int64), device=x.device)
                i2 = torch.as_tensor(np.linspace(0, x.shape[3], self.h, endpoint=False).astype(np.int64), device=x.device)
                x = x.index_select(2, i1).index_select(3, i2)
                insz = max(insz, self.h)
            elif self.resizing != 'interpolation':
                raise Exception(f'Wrong resizing method. It should be: interpolation or sampling. '
                                f'But the given value is {self.resizing}.')
        hist, totals = _RGBuvHistFunction.apply(x.contiguous(), insz, _HIST_METHODS[self.method], float(self.sigma),
                                                self._bins, bool(self.intensity_scale))
        if self.h < 64:
            hist = hist[:, :, :self.h, :self.h]
        if self.green_only:
            # The kernels normalise by T + EPS, T = the sum over all three pairings.  The G plane normalised by its
            # own un-normalised sum S_g is raw_g / (S_g + EPS) = hist_g / (sum(hist_g) + EPS / (T + EPS))
            # (losses.py:319-320); T is a constant for autograd, which touches the derivative only through that
            # EPS term (1e-6 / T, T in the thousands).
            g = hist[:, 1:2]
            hist = g / (g.sum(dim=(1, 2, 3), keepdim=True) + EPS / (totals.view(-1, 1, 1, 1) + EPS))
        return hist


class _HellingerSqFunction(torch.autograd.Function):
    """q = sum (sqrt(target_hist) - sqrt(input_hist))^2; gradient to input_hist only."""

    @staticmethod
    def forward(ctx, target_hist, input_hist):
        lib = _lib.load()
        out = torch.empty(1, dtype=torch.float32, device=input_hist.device)
        _lib.check(lib.st_hellinger_sq_fwd(_lib.dev_ptr(target_hist, "target_hist"), _lib.dev_ptr(input_hist, "input_hist"),
                                           input_hist.numel(), _lib.dev_ptr(_workspace(input_hist.device), "ws"),
                                           _lib.dev_ptr(out, "out"), _lib.stream_ptr()), "st_hellinger_sq_fwd")
        ctx.save_for_backward(target_hist, input_hist)
        return out.reshape(())

    @staticmethod
    def backward(ctx, gq):
        target_hist, input_hist = ctx.saved_tensors
        lib = _lib.load()
        dh = torch.empty_like(input_hist)
        gq = gq.contiguous().reshape(1)
        _lib.check(lib.st_hellinger_sq_bwd(_lib.dev_ptr(target_hist, "target_hist"), _lib.dev_ptr(input_hist, "input_hist"),
                                           _lib.dev_ptr(gq, "gq"), 1.0, _lib.dev_ptr(dh, "dh"), input_hist.numel(),
                                           _lib.stream_ptr()), "st_hellinger_sq_bwd")
        return None, dh


def hellinger_sq(target_hist, input_hist):
    return _HellingerSqFunction.apply(target_hist.detach().contiguous(), input_hist.contiguous())


class HistLoss(nn.Module):
    """losses.py:325-347.  ``reduce_q`` (optional) is applied to the scalar under the outer sqrt --
    data-parallel training passes an all-reduce there so the loss equals the single-GPU
    large-batch value (the sqrt is OUTSIDE the batch sum); ``global_batch`` then replaces B."""

    def __init__(self, intensity_scale=True, histogram_size=64, max_input_size=128, hist_boundary=[-3, 3],
                 method='inverse-quadratic'):
        super().__init__()
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.histogram_block = RGBuvHistBlock(insz=max_input_size, h=histogram_size, intensity_scale=intensity_scale,
                                              method=method, hist_boundary=list(hist_boundary), device=device)

    @staticmethod
    def q_from_hists(input_hist, target_hist):
        """The scalar under the outer sqrt of losses.py:342-344 for THIS process's images."""
        return hellinger_sq(target_hist, input_hist)

    @staticmethod
    def from_q(q, batch):
        """losses.py:342-347 from the (batch-wide) scalar under the sqrt."""
        histogram_loss = (1 / np.sqrt(2.0) * (torch.sqrt(q)) / batch)
        return histogram_loss * 100

    def from_hists(self, input_hist, target_hist, reduce_q=None, global_batch=None):
        q = self.q_from_hists(input_hist, target_hist)
        if reduce_q is not None:
            q = reduce_q(q)
        return self.from_q(q, input_hist.shape[0] if global_batch is None else global_batch)

    def forward(self, input, target, reduce_q=None, global_batch=None):
        input_hist = self.histogram_block(input)
        with torch.no_grad():
            target_hist = self.histogram_block(target)
        return self.from_hists(input_hist, target_hist, reduce_q, global_batch)


# --------------------------------------------------------------------------------------------
class _MomentsFunction(torch.autograd.Function):
    """calc_2_moments: (mu [b,c,1], cov [b,c,c] un-normalised), MFMA covariance both ways."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        b, c, h, w = (int(v) for v in x.shape)
        n = h * w
        mean, _ = ops.mean_std(x, 1e-5)
        mean = mean.reshape(b, c)
        cov = torch.empty((b, c, c), dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.st_cov_workspace_bytes(b, c, n) // 4, dtype=torch.float32, device=x.device)
        ev = None
        if ops.KERNEL_TIMER.enabled:
            ev = ops.KERNEL_TIMER.record("cov_fwd_kernel", 2.0 * b * c * c * n, 4.0 * x.numel())
        _lib.check(lib.st_cov_fwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(mean, "mean"), _lib.dev_ptr(cov, "cov"),
                                  _lib.dev_ptr(ws, "workspace"), b, c, n, _lib.stream_ptr()), "st_cov_fwd")
        if ev is not None:
            ev.record()
        ctx.save_for_backward(x, mean)
        return mean.reshape(b, c, 1), cov

    @staticmethod
    def backward(ctx, gmu, gcov):
        x, mean = ctx.saved_tensors
        lib = _lib.load()
        b, c, h, w = (int(v) for v in x.shape)
        dx = torch.empty_like(x)
        gcov = gcov.contiguous()
        gmu = gmu.contiguous().reshape(b, c) if gmu is not None else None
        _lib.check(lib.st_cov_bwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(mean, "mean"), _lib.dev_ptr(gcov, "gcov"),
                                  _lib.dev_ptr(gmu, "gmean") if gmu is not None else None, _lib.dev_ptr(dx, "dx"),
                                  b, c, h * w, _lib.stream_ptr()), "st_cov_bwd")
        return dx


class VincentStyleLoss(nn.Module):
    """losses.py:42-90.  The covariance (the O(C^2 HW) "Gram" term) and its gradient are MFMA kernels
    (csrc/cov.hip); the C x C eigendecompositions of ``l2wass_dist`` are rocSOLVER's syevd -- the driver
    ``torch.linalg.eigh`` uses -- called directly (``eigh.syevd``: no per-call host read of ``info``); the C x C
    products (round 3) are st_bmm_f32, a hand-written fp32 MFMA batched GEMM (csrc/bmm.hip).

    ``forward(input, target)`` is the reference's per-layer call.  A training step has five style layers and the
    decompositions are latency-bound (11.5 ms for [8, 512, 512], 13.0 ms for [16, 512, 512]), so the step engine
    uses the two-stage form instead:

    * ``prepare_targets(target_feats)`` -- everything that depends on the STYLE image only (mean, covariance,
      ``eigh(cov_stl)``, ``root_cov_stl``, its trace), layers with equal channel counts batched into one
      decomposition, each group on its own side stream: issued right after the style image's loss-VGG pass, it
      overlaps the encoder / decoder / loss-VGG convolutions that follow;
    * ``forward_multi(input_feats, prepared)`` -- the stylised side: covariances on the main stream, then
      ``root @ cov @ root`` and its eigenvalues per group, again concurrently.
    Same arithmetic per layer as ``forward``; the sum over layers is what trainer.py:257-263 computes."""

    def __init__(self, scale_factor=1e-5):
        super().__init__()
        warnings.warn("\n Using VincentStyleLoss might me slow!")
        self.scale_factor = scale_factor
        self._streams = {}

    @staticmethod
    def calc_2_moments(x):
        return _MomentsFunction.apply(x.contiguous())

    @staticmethod
    def matrix_diag(diagonal):
        return torch.diag_embed(diagonal)

    # ---- pieces of l2wass_dist (losses.py:67-81) ------------------------------------------------------
    #: how ``root_cov_stl = cov_stl^(1/2)`` (losses.py:69-71) is evaluated.  "eigh": the reference's route, V sqrt(clip(L,
    #: 0)) V^T from an eigendecomposition (rocSOLVER: 13 ms for the two 512-channel layers of a batch of 8, a chain of
    #: ~2000 dependent tiny launches).  "newton-schulz" (default): the same matrix function by the coupled
    #: Newton-Schulz iteration -- 3 batched C x C GEMMs per iteration on the matrix pipe, no eigendecomposition;
    #: measured on the loss-VGG features of the BASELINE configs[2] step it is as close to the float64 result as the
    #: fp32 eigh route is (both 1e-5 .. 4e-4 of the layer's distance; tests/test_gpu_backward.py checks both routes
    #: against the reference's golden vectors).  No gradient flows through this side (style image only).
    #: The iteration is used only when the layer has more pixels than channels (H*W > C: every BASELINE config; a
    #: covariance of fewer samples than channels is rank-deficient by construction, and on an exactly singular input
    #: the iteration's null directions are decided by rounding noise) -- otherwise the eigh route runs.
    #: Convergence, stated exactly: a normalised eigenvalue p << 1 grows by (3 - p)^2 / 4 ~ 2.25 per iteration, so
    #: after 16 iterations directions down to ~1e-5 of the (Frobenius) norm have converged; one at 1e-6 has reached
    #: ~0.66 sqrt(lambda), i.e. its part of the root is 3e-4 of the root's norm off (the fp32 eigh route: ~5e-5 there),
    #: and such a direction carries ~1e-6 of var_overlap.  A rounding-NEGATIVE eigenvalue of the PSD input at -1e-7 of
    #: the norm (what fp32 covariance rounding produces) has grown to -0.04 after 16 iterations and contributes
    #: -7e-5 of the norm to the root; at -1e-6 it would leave the basin two iterations later -- hence exactly 16, and
    #: tests/test_gpu_eigh.py::test_style_root_routes_on_ill_conditioned_covariance pins both routes against float64
    #: on a decaying spectrum with exact-zero rows (dead post-ReLU channels), var_overlap included.
    #: ST_STYLE_ROOT=eigh|newton-schulz overrides the default for A/B runs.
    STYLE_ROOT = os.environ.get("ST_STYLE_ROOT", "newton-schulz")
    NS_ITERS = 16
    #: the C x C products on st_bmm_f32 (csrc/bmm.hip, fp32 MFMA); ST_NO_HIP_BMM=1: torch.matmul / bmm (rocBLAS), for A/B
    HIP_BMM = os.environ.get("ST_NO_HIP_BMM") is None

    @classmethod
    def _style_side(cls, cov_stl, n_pixels=None):
        """(root_cov_stl, tr_cov_stl) of losses.py:69-72.  ``n_pixels``: H*W of the feature map behind ``cov_stl``."""
        if cls.STYLE_ROOT == "eigh" or n_pixels is None or n_pixels <= cov_stl.shape[-1]:
            from . import eigh
            eigvals, eigvects = eigh.syevd(cov_stl)
            pos = eigvals.clip(0)
            scaled = eigvects * torch.sqrt(pos).unsqueeze(-2)
            if ops.bmm_ok(scaled) and cls.HIP_BMM:
                root_cov_stl = ops.bmm_raw(scaled, eigvects, False, True)
            else:
                root_cov_stl = torch.matmul(scaled, torch.transpose(eigvects, -1, -2))
            return root_cov_stl, torch.sum(pos, dim=1, keepdim=True)
        # cov_stl = (x - mu)(x - mu)^T is positive semi-definite: sum(clip(eigenvalues, 0)) is its trace (the argument
        # of _synth_side for tr_cov_synth), and V sqrt(clip(L, 0)) V^T is its principal square root.
        tr_cov_stl = torch.diagonal(cov_stl, dim1=-2, dim2=-1).sum(dim=1, keepdim=True)
        n = cov_stl.shape[-1]
        eye = torch.eye(n, dtype=cov_stl.dtype, device=cov_stl.device).expand_as(cov_stl)
        norm = cov_stl.flatten(1).norm(dim=1).clamp_min(1e-30).view(-1, 1, 1)      # Frobenius >= spectral norm
        y = cov_stl / norm                                                          # spectrum in [0, 1]
        z = eye
        eye15 = 1.5 * eye
        hip = ops.bmm_ok(y) and cls.HIP_BMM
        for _ in range(cls.NS_ITERS):                     # Y -> A^(1/2), Z -> A^(-1/2)
            if hip:                                       # T = 1.5 I - 0.5 Z Y in the GEMM's epilogue
                t = ops.bmm_raw(z.contiguous(), y, alpha=-0.5, diag=1.5)
                y, z = ops.bmm_raw(y, t), ops.bmm_raw(t, z.contiguous())
            else:
                t = torch.baddbmm(eye15, z, y, beta=1.0, alpha=-0.5)
                y, z = torch.bmm(y, t), torch.bmm(t, z)
        return y * torch.sqrt(norm), tr_cov_stl

    @staticmethod
    def _synth_side(root_cov_stl, tr_cov_stl, cov_synth):
        """tr_cov_stl + tr_cov_synth - 2 var_overlap per matrix (losses.py:73-80 without the mean term)."""
        from . import eigh
        # losses.py:73 sums eigvalsh(cov_synth).clip(0).  cov_synth = (x - mu)(x - mu)^T is positive semi-definite by
        # construction, so that sum IS its trace up to fp32 eigenvalue rounding (<= C * 1e-7 relative, the slightly
        # negative computed eigenvalues the clip removes); its gradient, the projector onto range(cov_synth), acts
        # on x - mu exactly like the identity.  Taking the trace saves one of the three batched
        # eigendecompositions per layer.
        tr_cov_synth = torch.diagonal(cov_synth, dim1=-2, dim2=-1).sum(dim=1, keepdim=True)
        if ops.bmm_ok(cov_synth) and VincentStyleLoss.HIP_BMM and root_cov_stl.shape == cov_synth.shape:
            cov_prod = ops.bmm(ops.bmm(root_cov_stl, cov_synth), root_cov_stl)
        else:
            cov_prod = torch.matmul(torch.matmul(root_cov_stl, cov_synth), root_cov_stl)
        var_overlap = torch.sum(torch.sqrt(eigh.eigvalsh(cov_prod).clip(0.1)), dim=1, keepdim=True)
        return tr_cov_stl + tr_cov_synth - 2 * var_overlap

    def l2wass_dist(self, mean_stl, cov_stl, mean_synth, cov_synth, n_pixels=None):
        root_cov_stl, tr_cov_stl = self._style_side(cov_stl, n_pixels)
        mean_diff_squared = torch.mean((mean_synth - mean_stl) ** 2)
        return mean_diff_squared + self._synth_side(root_cov_stl, tr_cov_stl, cov_synth)

    def forward(self, input_f, target_f):
        mean_synth, cov_synth = self.calc_2_moments(input_f)
        with torch.no_grad():
            mean_stl, cov_stl = self.calc_2_moments(target_f)
        loss = self.l2wass_dist(mean_stl, cov_stl, mean_synth, cov_synth, int(target_f.shape[2] * target_f.shape[3]))
        return loss.mean() * self.scale_factor

    # ---- the step engine's form: layers grouped by channel count, groups on side streams ---------------
    def _side_stream(self, key):
        """Side stream number ``key`` (0 .. LANES-1) of the current device: streams that were MEASURED to overlap with
        the current stream and with each other (streams.concurrent_lanes), not merely distinct stream objects."""
        return self._lanes()[key]

    def _lanes(self):
        from .streams import concurrent_lanes
        return concurrent_lanes(self.LANES + 1)       # (high stream priority for the eigen lanes: measured, no effect)

    def prefetch_stream(self):
        """The one measured-concurrent side stream the eigen jobs do NOT use (main + LANES + this one = the four
        hardware queues): Trainer runs the next batch's weight-independent forward on it."""
        return self._lanes()[self.LANES]

    @staticmethod
    def _groups(feats):
        groups = {}
        for i, f in enumerate(feats):
            groups.setdefault(int(f.shape[1]), []).append(i)
        return sorted(groups.items(), key=lambda kv: -kv[0])          # widest (slowest) group first

    #: Side streams ("lanes") for the per-group eigen / Newton-Schulz jobs.  The GPU has FOUR hardware queues per
    #: process (GPU_MAX_HW_QUEUES): streams beyond that share a queue and run back to back in submission order.
    #: Round 2 first used one stream per (side, channel count) = 8 side streams: 54.7 ms per Gram step.  With three
    #: lanes in all -- the widest group (the two 512-channel taps, stacked into one 13 ms call) alone on lane 0, the
    #: next group on lane 1, every narrower group back to back on lane 2 -- the step takes 47.0 ms, provided the lanes
    #: sit on distinct hardware queues (streams.py picks them by measurement; 52-54 ms when two of them or a lane
    #: and the main stream share a queue).  Splitting the widest group's two layers over two lanes instead of
    #: stacking them: 70.5 ms (two rocSOLVER chains of ~2000 dependent launches each slow each other down far more
    #: than stacking costs).
    #: Two lanes for the eigen jobs (47.4 ms; three: 47.0), so that a fourth queue stays free for prefetch_stream().
    LANES = int(os.environ.get("ST_VINCENT_LANES", "2"))

    @torch.no_grad()
    def prepare_targets(self, target_feats, inline: bool = False):
        """Style-image side of every layer; returns an opaque list for ``forward_multi``.  Work is issued on side
        streams and NOT waited for here (one stacked call per group; lanes as in forward_multi).  ``inline``: on the
        CURRENT stream instead (a caller that is itself a side stream: the trainer's prefetch of the next batch)."""
        cur = torch.cuda.current_stream()
        prepared = []
        for g, (c, idxs) in enumerate(self._groups(target_feats)):
            st = cur if inline else self._side_stream(min(g, self.LANES - 1))
            if st is not cur:
                st.wait_stream(cur)
            with torch.cuda.stream(st):
                moments = [self.calc_2_moments(target_feats[i]) for i in idxs]
                mean_stl = [m for m, _ in moments]
                # layers of one group may differ in H*W (relu4_1 / conv4_4 do not): the smallest decides the route
                n_pixels = min(int(target_feats[i].shape[2] * target_feats[i].shape[3]) for i in idxs)
                root, tr = self._style_side(torch.cat([cv for _, cv in moments]), n_pixels)
            for i in idxs:
                target_feats[i].record_stream(st)
            prepared.append((c, idxs, mean_stl, root, tr, st))
        return prepared

    def forward_multi(self, input_feats, prepared, on_jobs_issued=None):
        """Sum over the layers of ``forward(input_feats[i], target_i)`` given ``prepare_targets(targets)``.
        ``on_jobs_issued``: called after every eigen job has been issued on its lane and BEFORE the current stream is
        made to wait for them -- the place to put independent work on another stream (Trainer's prefetch)."""
        cur = torch.cuda.current_stream()
        moments = [self.calc_2_moments(f) for f in input_feats]          # MFMA covariances, main stream
        nb = int(input_feats[0].shape[0])
        parts, used = [], []
        for g, (c, idxs, mean_stl, root, tr, st_stl) in enumerate(prepared):      # widest group first
            lane = min(g, self.LANES - 1)
            st = self._side_stream(lane)
            if st not in used:
                st.wait_stream(cur)
                used.append(st)
            st.wait_stream(st_stl)
            with torch.cuda.stream(st):
                cov_synth = torch.cat([moments[i][1] for i in idxs]) if len(idxs) > 1 else moments[idxs[0]][1]
                rest = self._synth_side(root, tr, cov_synth)
                for k, i in enumerate(idxs):
                    # the mean term is per layer (a scalar over the layer's whole batch, losses.py:75)
                    dist = ((moments[i][0] - mean_stl[k]) ** 2).mean() + rest[k * nb:(k + 1) * nb]
                    parts.append((dist.mean() * self.scale_factor, st))
            for t in (root, tr, *mean_stl, *(moments[i][k] for i in idxs for k in (0, 1))):
                t.record_stream(st)
        if on_jobs_issued is not None:
            on_jobs_issued()
        total = None
        for val, st in parts:
            cur.wait_stream(st)
            val.record_stream(cur)
            total = val if total is None else total + val
        return total
