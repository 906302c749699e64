"""AdaIN with the reference's surface (src/models/transformer.py:7-59) on HIP kernels."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .autograd import _needs_grad


class AdaIN(nn.Module):
    def __init__(self, eps: float = 1e-5):
        super().__init__()
        self.eps_limit = eps
        self.content_eps = nn.Parameter(torch.FloatTensor([1e-5]), requires_grad=True)  # transformer.py:11
        self.style_eps = nn.Parameter(torch.FloatTensor([1e-5]), requires_grad=True)    # transformer.py:12

    @staticmethod
    def calc_mean_std(feat, eps=1e-5):
        """transformer.py:15-24 (one HIP wavefront-reduction launch instead of var + mean passes)."""
        size = feat.size()
        assert (len(size) == 4)
        assert 0 < eps < 1.0
        return ops.mean_std(feat, float(eps))

    @staticmethod
    def adjust_sigmoid(x, upper_limit=0.3):
        """transformer.py:27-36: sqrt(|x|) clamped to [1e-5, upper_limit]; clamped branches are
        Python floats.  Like the reference this reads the 1-element tensor on the host."""
        lower_limit = 1e-5
        x = torch.pow(torch.abs(x), 0.5)
        if x < lower_limit:
            scaled_output = lower_limit
        elif x > upper_limit:
            scaled_output = upper_limit
        else:
            scaled_output = x
        return scaled_output

    @staticmethod
    def adjust_sigmoid_device(x, upper_limit=0.3):
        """adjust_sigmoid without the host read: sqrt(|x|).clamp(1e-5, upper_limit) stays a 1-element device
        tensor.  Same value as transformer.py:27-36 in all three branches; where the reference returns a Python
        float (clamped, no gradient) the clamp passes a ZERO gradient to the parameter instead of none."""
        return torch.pow(torch.abs(x), 0.5).clamp(1e-5, upper_limit)

    def clamped_masks(self):
        """Per eps parameter, a 1-element device bool: True where adjust_sigmoid takes one of its two constant
        branches.  There the reference hands the optimizer NO gradient (a Python float carries no graph), so the
        parameter is not touched at all -- no update, no weight decay, no momentum drift; the trainer uses these
        masks to restore a clamped parameter after ``optimizer.step()`` (the device-side clamp delivers a zero
        gradient instead of none)."""
        out = []
        for p in (self.content_eps, self.style_eps):
            x = torch.pow(torch.abs(p.detach()), 0.5)
            out.append((x < 1e-5) | (x > self.eps_limit))
        return out

    def get_current_eps(self):
        """transformer.py:38-44."""
        if torch.cuda.is_available():
            return [float(self.adjust_sigmoid(self.style_eps, self.eps_limit)),
                    float(self.adjust_sigmoid(self.content_eps, self.eps_limit))]
        return [self.adjust_sigmoid(self.style_eps, self.eps_limit),
                self.adjust_sigmoid(self.content_eps, self.eps_limit)]

    def adaptive_instance_normalization(self, content_feat, style_feat, style_eps=1e-5, content_eps=1e-5,
                                        alpha: float = 1.0):
        """transformer.py:46-54, with the alpha blend of trainer.py:434/702 fused (alpha=1: identity)."""
        assert (content_feat.size()[:2] == style_feat.size()[:2])
        if _needs_grad(content_feat, style_feat) or any(
                isinstance(e, torch.Tensor) and _needs_grad(e) for e in (style_eps, content_eps)):
            from .autograd_adain import adain_autograd
            return adain_autograd(content_feat, style_feat, content_eps, style_eps, alpha)
        if isinstance(content_eps, torch.Tensor) and isinstance(style_eps, torch.Tensor) and content_eps.is_cuda:
            eps2 = torch.cat([content_eps.detach().reshape(1), style_eps.detach().reshape(1)])
            out, _ = ops.adain(content_feat, style_feat, 0.0, 0.0, alpha, eps_dev=eps2)
            return out
        out, _ = ops.adain(content_feat, style_feat, float(content_eps), float(style_eps), alpha)
        return out

    def forward(self, content_feat, style_feat, alpha: float = 1.0):
        """transformer.py:56-59.  ``alpha`` (default 1.0 = reference behaviour) fuses
        ``alpha * t + (1 - alpha) * content_feat`` into the same kernel.  The two eps stay on the device
        (adjust_sigmoid_device): no host synchronisation inside a step."""
        return self.adaptive_instance_normalization(
            content_feat, style_feat,
            style_eps=self.adjust_sigmoid_device(self.style_eps, self.eps_limit),
            content_eps=self.adjust_sigmoid_device(self.content_eps, self.eps_limit),
            alpha=alpha)
