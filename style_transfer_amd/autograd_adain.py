"""AdaIN with gradients: HIP forward (stats saved) + HIP backward (transformer.py:46-59)."""
from __future__ import annotations

import torch

from . import _lib, ops


class _AdaINFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, content, style, eps2, alpha):
        # eps2: 2-element device tensor (eps_content, eps_style), possibly with grad (sqrt(|param|) of
        # adjust_sigmoid); it is read by the kernel, never by the host
        out, stats = ops.adain(content, style, 0.0, 0.0, alpha, save_stats=True, eps_dev=eps2.detach().contiguous())
        ctx.alpha = alpha
        ctx.save_for_backward(content, style, stats)
        return out

    @staticmethod
    def backward(ctx, gout):
        content, style, stats = ctx.saved_tensors
        lib = _lib.load()
        gout = gout.contiguous()
        n, c, hc, wc = (int(v) for v in content.shape)
        hs, ws = int(style.shape[2]), int(style.shape[3])
        dcontent = torch.empty_like(content) if ctx.needs_input_grad[0] else None
        dstyle = torch.empty_like(style) if ctx.needs_input_grad[1] else None
        deps = torch.empty(2, dtype=torch.float32, device=content.device)
        work = torch.empty(n * c * 2, dtype=torch.float32, device=content.device)
        _lib.check(lib.st_adain_bwd(_lib.dev_ptr(gout, "gout"), _lib.dev_ptr(content, "content"),
                                    _lib.dev_ptr(style, "style"), _lib.dev_ptr(stats, "stats"), float(ctx.alpha),
                                    _lib.dev_ptr(dcontent, "dcontent") if dcontent is not None else None,
                                    _lib.dev_ptr(dstyle, "dstyle") if dstyle is not None else None,
                                    _lib.dev_ptr(deps, "deps"), _lib.dev_ptr(work, "workspace"),
                                    n, c, hc * wc, hs * ws, _lib.stream_ptr()), "st_adain_bwd")
        return dcontent, dstyle, (deps if ctx.needs_input_grad[2] else None), None


def adain_autograd(content, style, content_eps, style_eps, alpha: float = 1.0):
    """content_eps / style_eps: Python floats (clamped branches of adjust_sigmoid, no grad) or
    1-element tensors that carry the graph back to the trainable eps parameters."""
    dev = content.device

    def as_t(e):
        if isinstance(e, torch.Tensor):
            return e.reshape(1)
        return torch.tensor([float(e)], dtype=torch.float32, device=dev)

    return _AdaINFunction.apply(content, style, torch.cat([as_t(content_eps), as_t(style_eps)]), float(alpha))
