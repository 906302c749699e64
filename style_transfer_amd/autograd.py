"""torch.autograd glue: each HIP forward op paired with its HIP backward ops.

Forward-only (no input or parameter requires grad, or grad mode is off) calls go
straight to ``ops`` without building an autograd node.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


def _needs_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def conv3x3_apply(x, weight, bias, cache, key, upsample2x: bool, pad_mode: int, relu: bool):
    """Fused [nearest x2 upsample] -> pad 1 -> Conv2d(3x3) -> [ReLU]."""
    cout = int(weight.shape[0])
    if not _needs_grad(x, weight, bias):
        wp = cache.get(key, weight, False, forward_only=True,
                       upsample2x=upsample2x and ops._pad_code(pad_mode) == ops.PAD_REPLICATE)
        return ops.conv3x3(x, wp, bias.detach() if bias is not None else None, cout, upsample2x, pad_mode, relu)
    from .autograd_conv import Conv3x3Function
    return Conv3x3Function.apply(x, weight, bias, cache, key, upsample2x, pad_mode, relu)


def maxpool2x2_apply(x, ceil_mode: bool):
    if not _needs_grad(x):
        return ops.maxpool2x2(x, ceil_mode)
    from .autograd_conv import MaxPool2x2Function
    return MaxPool2x2Function.apply(x, ceil_mode)


def conv1x1_apply(x, weight, bias):
    if not _needs_grad(x, weight, bias):
        return ops.conv1x1_3to3(x, weight, bias)
    raise NotImplementedError("gradients through the encoder's 1x1 input conv are not part of the hot path "
                              "(the encoder is frozen, trainer.py:286-297)")


def sympad_apply(x, padding):
    if not _needs_grad(x):
        return ops.sympad(x, padding)
    raise NotImplementedError("SymmetricPadding2D backward: use the fused conv path")
