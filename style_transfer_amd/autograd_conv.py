"""autograd Functions for the fused conv layer and the max pool (HIP forward + HIP backward)."""
from __future__ import annotations

import os

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
        # (fp32 mode, replicate-padded layer behind the x2 upsample: sub-pixel forward, see DecoderChainFunction)
        wp = cache.get(key, weight, False, upsample2x=bool(upsample2x) and ops._pad_code(pad_mode) == ops.PAD_REPLICATE)
        y = ops.conv3x3(x, wp, bias.detach() if bias is not None else None, cout, upsample2x, pad_mode, relu)
        ctx.cache, ctx.key = cache, key
        fam = ops.conv_family(wp, int(weight.shape[0]), int(weight.shape[1]))           # backward uses the forward's precision
        ctx.compute = "f32" if fam == "f32up2" else fam
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


class DecoderChainFunction(torch.autograd.Function):
    """The whole decoder stack (generator.py:85-130) as ONE autograd node.

    Every intermediate activation has exactly one consumer (the next conv), so the ReLU backward of layer
    i-1 can be fused into the data-gradient launch of layer i (``dx_gate``): the gradient that reaches layer
    i-1 is already gated, and its own data-gradient and weight-gradient launches read ``dy`` alone instead of
    ``dy`` and the saved activation -- one full tensor read less in each of them.

    ``specs``: per layer (key, cout, upsample2x, relu); ``params``: w0, b0, w1, b1, ...

    ``cache.direct_grads`` (set by the trainer's flat gradient bucket, trainer.GradBucket): the weight-gradient
    kernel writes straight into the ``.grad`` the bucket pre-allocated for each parameter (a slice of ONE flat
    buffer, all-reduced in place) and the node returns no gradient for it -- no per-parameter accumulate or copy
    kernels.  The write OVERWRITES: valid because the trainer runs the decoder once per backward.
    """

    #: test hook: a list to receive the layer outputs of the next forward (tests/test_gpu_parity_step.py reads the
    #: ReLU masks the step actually took); None in every other use
    TAP = None

    @staticmethod
    def forward(ctx, x, cache, specs, *params):
        acts = [x]
        compute = []        # per layer: the backward of a layer uses its forward's precision
        for i, (key, cout, up, relu) in enumerate(specs):
            w, b = params[2 * i], params[2 * i + 1]
            # fp32 mode: a layer behind the x2 upsample runs its FORWARD in the sub-pixel form (16 instead of 36
            # multiply-adds per source pixel and channel pair, the same products regrouped; generator.py:92-93,
            # 105-106, 112-113) under autograd as well; its data and weight gradients are the direct form's (the
            # adjoint of the same linear map).  Whole-step gradient parity no longer depends on which last-bit
            # rounding the forward has: tests/test_gpu_parity_step.py pins the switches instead.
            wp = cache.get(key, w, False, upsample2x=up)
            fam = ops.conv_family(wp, int(w.shape[0]), int(w.shape[1]))
            compute.append("f32" if fam == "f32up2" else fam)
            acts.append(ops.conv3x3(acts[-1], wp, b.detach(), cout, up, ops.PAD_REPLICATE, relu))
        ctx.cache, ctx.specs, ctx.compute = cache, specs, compute
        ctx.save_for_backward(*acts[:-1], *params)          # layer inputs (the last output is not needed)
        if DecoderChainFunction.TAP is not None:
            DecoderChainFunction.TAP.extend(a.detach() for a in acts[1:])
        return acts[-1]

    @staticmethod
    def backward(ctx, g):
        n = len(ctx.specs)
        saved = ctx.saved_tensors
        acts, params = saved[:n], saved[n:]
        g = g.contiguous()
        grads = [None] * (2 * n)
        dx = None
        for i in range(n - 1, -1, -1):
            key, cout, up, relu = ctx.specs[i]
            w, b = params[2 * i], params[2 * i + 1]
            x_i = acts[i]
            # g = d loss / d (output of layer i); already ReLU-gated by the launch that produced it,
            # except at the top of the chain, where the layer's own output would be needed -- the
            # reference's last decoder conv has no ReLU (generator.py:117), which is asserted in Decoder
            if ctx.needs_input_grad[3 + 2 * i] or ctx.needs_input_grad[4 + 2 * i]:
                direct = (getattr(ctx.cache, "direct_grads", False) and ctx.needs_input_grad[3 + 2 * i]
                          and ctx.needs_input_grad[4 + 2 * i] and w.grad is not None and b.grad is not None
                          and w.grad.is_contiguous() and b.grad.is_contiguous())
                dw, db = ops.conv3x3_wgrad(x_i, g, None, want_bias=True, upsample2x=up,
                                           pad_mode=ops.PAD_REPLICATE, compute=ctx.compute[i],
                                           dw_out=w.grad if direct else None, db_out=b.grad if direct else None)
                if not direct:
                    grads[2 * i] = dw if ctx.needs_input_grad[3 + 2 * i] else None
                    grads[2 * i + 1] = db if ctx.needs_input_grad[4 + 2 * i] else None
            if i > 0 or ctx.needs_input_grad[0]:
                wpt = ctx.cache.get(key, w, True, ctx.compute[i])
                gate = x_i if (i > 0 and ctx.specs[i - 1][3]) else None     # input is a ReLU output
                g = ops.conv3x3_dgrad(g, None, wpt, int(w.shape[1]), up, ops.PAD_REPLICATE, dx_gate=gate)
                if i == 0:
                    dx = g
        return (dx, None, None, *grads)


class DecoderChainG8Function(torch.autograd.Function):
    """DecoderChainFunction for the bf16 compute mode with G8 hand-overs in BOTH directions.

    Forward: generator._g8_chain's launches (fp32 in, G8 between ALL the layers, bit-identical to the fp32-activation
    kernels; the 3-channel top layer's forward, its data gradient's gate and its vector-ALU weight gradient read the
    G8 form too -- TOP_G8 = False keeps round 2's fp32 hand-over to the top layer for A/B).  The saved layer inputs
    are the G8 tensors: half the bytes, and already the bf16 values the weight gradient rounds its input to.

    Backward, per layer from the top: ONE data-gradient call (ops.conv3x3_dgrad_g8_chain) takes the gated gradient
    in G8 (staged by LDS-DMA), folds the padded-domain result, applies the ReLU gate of the layer below from its G8
    activation and writes the gradient in the two bf16 forms its consumers take -- G8 for the next data-gradient
    launch, pixel-major bf16 plus bias partial sums for the next weight-gradient GEMM (ops.conv3x3_wgrad_pre): no
    fp32 gradient tensor between the layers, no weight-gradient prep pass over dy.  Same fp32 values per element
    as DecoderChainFunction before the (identical) bf16 roundings; the bias gradient is summed in another (fixed)
    order.  Shapes outside the staged weight-gradient form use DecoderChainFunction (Decoder.forward decides).
    """

    TOP_G8 = os.environ.get("ST_DEC_TOP_F32") is None

    @staticmethod
    def usable(x, specs, params) -> bool:
        n, c, h, w = (int(v) for v in x.shape)
        for i, (key, cout, up, relu) in enumerate(specs):
            cin = int(params[2 * i].shape[1])
            if up:
                h, w = 2 * h, 2 * w
            if cin % 16 or w % 32 or h % 4:
                return False
            if cout != 3 and (cout % 8 or not ops.wgrad_pre_ok(n, cout, cin, h, w)):
                return False
        return int(specs[-1][1]) == 3 and not specs[-1][3] and all(sp[3] for sp in specs[:-1])

    @staticmethod
    def forward(ctx, x, cache, specs, *params):
        n_l = len(specs)
        acts = [x]
        for i, (key, cout, up, relu) in enumerate(specs):
            w, b = params[2 * i], params[2 * i + 1]
            wp = cache.get(key, w, False, "bf16")
            acts.append(ops.conv3x3_g8(acts[-1], wp, b.detach(), cout, upsample2x=up, pad_mode=ops.PAD_REPLICATE,
                                       relu=relu, pool=False,
                                       y_g8=i < n_l - (1 if DecoderChainG8Function.TOP_G8 else 2)))
        ctx.cache, ctx.specs = cache, specs
        ctx.save_for_backward(*acts[:-1], *params)
        if DecoderChainFunction.TAP is not None:            # test hook: every layer output, as fp32 NCHW
            DecoderChainFunction.TAP.extend(ops.from_g8(a) if ops.is_g8(a) else a.detach() for a in acts[1:])
        return acts[-1]

    @staticmethod
    def backward(ctx, g):
        n = len(ctx.specs)
        saved = ctx.saved_tensors
        acts, params = saved[:n], saved[n:]
        g = g.contiguous()
        grads = [None] * (2 * n)
        g16 = bias = None              # the bf16 operand forms of the gradient w.r.t. layer i's output (i < n - 1)
        dx = None
        for i in range(n - 1, -1, -1):
            key, cout, up, relu = ctx.specs[i]
            w, b = params[2 * i], params[2 * i + 1]
            x_i = acts[i]
            need_w, need_b = ctx.needs_input_grad[3 + 2 * i], ctx.needs_input_grad[4 + 2 * i]
            if need_w or need_b:
                direct = (getattr(ctx.cache, "direct_grads", False) and need_w and need_b and w.grad is not None
                          and b.grad is not None and w.grad.is_contiguous() and b.grad.is_contiguous())
                if i == n - 1 and ops.is_g8(x_i):         # 64 -> 3: fp32 dy, G8 x, the vector-ALU kernel
                    dw, db = ops.conv3x3_wgrad_cout3_g8(x_i, g, upsample2x=up, pad_mode=ops.PAD_REPLICATE,
                                                        dw_out=w.grad if direct else None,
                                                        db_out=b.grad if direct else None)
                elif i == n - 1:       # ... fp32 x (TOP_G8 = False)
                    dw, db = ops.conv3x3_wgrad(x_i, g, None, want_bias=True, upsample2x=up,
                                               pad_mode=ops.PAD_REPLICATE, compute="bf16",
                                               dw_out=w.grad if direct else None, db_out=b.grad if direct else None)
                else:
                    dw, db = ops.conv3x3_wgrad_pre(x_i, g16, bias, up, ops.PAD_REPLICATE,
                                                   dw_out=w.grad if direct else None,
                                                   db_out=b.grad if direct else None)
                if not direct:
                    grads[2 * i] = dw if need_w else None
                    grads[2 * i + 1] = db if need_b else None
            if i > 0 or ctx.needs_input_grad[0]:
                wpt = ctx.cache.get(key, w, True, "bf16")
                gate = x_i if i > 0 else None                   # every layer below the top one ends in a ReLU
                below_trains = i > 0 and (ctx.needs_input_grad[1 + 2 * i] or ctx.needs_input_grad[2 + 2 * i])
                dxf, g8, g16, bias = ops.conv3x3_dgrad_g8_chain(
                    g, wpt, int(w.shape[1]), up, gate, want_f32=i == 0, want_g8=i > 0, want_g16=below_trains,
                    want_bias=below_trains)
                g = g8
                if i == 0:
                    dx = dxf
        return (dx, None, None, *grads)
