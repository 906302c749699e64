"""Loss modules with the reference's names and constructor defaults (src/models/losses.py),
forward AND backward on HIP kernels.  Gradients flow to the first argument (``input`` /
``image``); the second (``target``) is treated as a constant, which is all the reference's
training step needs (targets are features of the un-trained content/style images,
trainer.py:446-457)."""
from __future__ import annotations

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


class AdaINStyleLoss(nn.Module):
    """losses.py:14-39: MSE of channel means + MSE of channel stds (unbiased var + 1e-5)."""

    @staticmethod
    def calc_mean_std(feat, eps=1e-5):
        size = feat.size()
        assert (len(size) == 4)
        return ops.mean_std(feat, float(eps))

    def calc_style_loss(self, input, target):
        assert (input.size() == target.size())
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
        return hist

    @staticmethod
    def backward(ctx, ghist):
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
    """losses.py:122-322.  Supported on the HIP path: h=64, resizing='interpolation',
    green_only=False (the configuration HistLoss builds, losses.py:333-336); other settings raise."""

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
        if h != 64 or green_only or resizing != 'interpolation':
            raise NotImplementedError("HIP RGBuvHistBlock supports h=64, green_only=False, resizing='interpolation'")
        if method not in _HIST_METHODS:
            raise Exception(f'Wrong kernel method. It should be either thresholding, RBF,'
                            f' inverse-quadratic. But the given value is {self.method}.')
        # float64 bin centres exactly as the reference builds them (losses.py:211-213)
        self._bins = [float(v) for v in np.linspace(self.hist_boundary[0], self.hist_boundary[1], num=self.h)]

    def forward(self, x):
        return _RGBuvHistFunction.apply(x.contiguous(), int(self.insz), _HIST_METHODS[self.method], float(self.sigma),
                                        self._bins, bool(self.intensity_scale))


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
    """losses.py:42-90.  The covariance (the O(C^2 HW) "Gram" term) and its gradient are MFMA
    kernels; the C x C eigendecompositions and products of ``l2wass_dist`` stay on
    ``torch.linalg.eigh`` / ``torch.matmul`` (rocSOLVER / rocBLAS), as SURVEY.md K7b records."""

    def __init__(self, scale_factor=1e-5):
        super().__init__()
        warnings.warn("\n Using VincentStyleLoss might me slow!")
        self.scale_factor = scale_factor

    @staticmethod
    def calc_2_moments(x):
        return _MomentsFunction.apply(x.contiguous())

    @staticmethod
    def matrix_diag(diagonal):
        return torch.diag_embed(diagonal)

    def l2wass_dist(self, mean_stl, cov_stl, mean_synth, cov_synth):
        eigvals, eigvects = torch.linalg.eigh(cov_stl)
        eigroot_mat = self.matrix_diag(torch.sqrt(eigvals.clip(0)))
        root_cov_stl = torch.matmul(torch.matmul(eigvects, eigroot_mat), torch.transpose(eigvects, -1, -2))
        tr_cov_stl = torch.sum(eigvals.clip(0), dim=1, keepdim=True)
        # losses.py:73 sums eigvalsh(cov_synth).clip(0).  cov_synth = (x - mu)(x - mu)^T is positive semi-definite by
        # construction, so that sum IS its trace up to fp32 eigenvalue rounding (<= C * 1e-7 relative, the slightly
        # negative computed eigenvalues the clip removes); its gradient, the projector onto range(cov_synth), acts
        # on x - mu exactly like the identity.  Taking the trace saves one of the three batched
        # eigendecompositions per layer (rocSOLVER: 12 ms for [8, 512, 512]).
        tr_cov_synth = torch.diagonal(cov_synth, dim1=-2, dim2=-1).sum(dim=1, keepdim=True)
        mean_diff_squared = torch.mean((mean_synth - mean_stl) ** 2)
        cov_prod = torch.matmul(torch.matmul(root_cov_stl, cov_synth), root_cov_stl)
        var_overlap = torch.sum(torch.sqrt(torch.linalg.eigvalsh(cov_prod).clip(0.1)), dim=1, keepdim=True)
        dist = mean_diff_squared + tr_cov_stl + tr_cov_synth - 2 * var_overlap
        return dist

    def forward(self, input_f, target_f):
        mean_synth, cov_synth = self.calc_2_moments(input_f)
        with torch.no_grad():
            mean_stl, cov_stl = self.calc_2_moments(target_f)
        loss = self.l2wass_dist(mean_stl, cov_stl, mean_synth, cov_synth)
        return loss.mean() * self.scale_factor
