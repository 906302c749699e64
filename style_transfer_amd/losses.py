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
