"""autograd Functions for the fused conv layer and the max pool (HIP forward + HIP backward)."""
from __future__ import annotations

import torch

from . import ops


class Conv3x3Function(torch.autograd.Function):
    """[nearest x2 upsample] -> pad 1 -> Conv2d(3x3) -> [ReLU], one MFMA launch each way.

    backward: data gradient = the same implicit-GEMM kernel on the rotated/transposed filter
    (ReLU gate applied while staging dy); weight/bias gradient = pixel-reducing MFMA GEMM.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, cache, key, upsample2x, pad_mode, relu):
        cout = int(weight.shape[0])
        wp = cache.get(key, weight, False)
        y = ops.conv3x3(x, wp, bias.detach() if bias is not None else None, cout, upsample2x, pad_mode, relu)
        ctx.cache, ctx.key = cache, key
        ctx.compute = "bf16" if wp.dtype == torch.bfloat16 else "f32"   # backward uses the forward's precision
        ctx.upsample2x, ctx.pad_mode, ctx.relu = upsample2x, pad_mode, relu
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wpt = ctx.cache.get(ctx.key, weight, True, ctx.compute)
            dx = ops.conv3x3_dgrad(dy, y, wpt, int(weight.shape[1]), ctx.upsample2x, ctx.pad_mode)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = ops.conv3x3_wgrad(x, dy, y, want_bias=ctx.has_bias, upsample2x=ctx.upsample2x,
                                       pad_mode=ctx.pad_mode, compute=ctx.compute)
            if not ctx.needs_input_grad[1]:
                dw = None
            if not (ctx.has_bias and ctx.needs_input_grad[2]):
                db = None
        return dx, dw, db, None, None, None, None, None


class MaxPool2x2Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ceil_mode):
        ctx.ceil_mode = ceil_mode
        ctx.save_for_backward(x)
        return ops.maxpool2x2(x, ceil_mode)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool2x2_bwd(x, dy.contiguous(), ctx.ceil_mode), None
