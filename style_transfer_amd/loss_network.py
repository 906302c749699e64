"""The perceptual loss network of the training step: torchvision ``vgg19().features`` truncated
at the last requested layer, every Conv2d under weight_norm, read through an
IntermediateLayerGetter (src/models/trainer.py:165-199, used at :446-457), on HIP kernels.

torchvision is not a dependency here: the VGG19 "E" layout is restated (3x3 convs with zero
padding 1, ReLU(inplace=True), floor-mode 2x2 max pools; indices 0..36).  Each
Conv2d -> ReLU pair is one fused MFMA launch.  Because torchvision's ReLUs are in-place, the
tensor a caller captured at a conv index has been overwritten by the following ReLU by the time
``forward`` returns -- unless that conv is the last layer kept.  ``forward`` reproduces exactly
that: captured conv outputs are post-ReLU, except when the conv is the final layer.
"""
from __future__ import annotations

import math
import os
import warnings
from collections import OrderedDict
from typing import Dict, List, Sequence, Union

import torch
import torch.nn as nn

from . import ops, precision
from .autograd import _needs_grad, conv3x3_apply, maxpool2x2_apply
from .generator import _PackedWeightCache

VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def vgg19_layers():
    """[(kind, cin, cout)] for torchvision vgg19().features indices 0..36."""
    layers, cin = [], 3
    for v in VGG19_CFG:
        if v == "M":
            layers.append(("pool", 0, 0))
        else:
            layers.append(("conv", cin, int(v)))
            layers.append(("relu", 0, 0))
            cin = int(v)
    return layers


class _NormalizeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mean, std):
        ctx.std = std
        return ops.normalize(x, mean, std)

    @staticmethod
    def backward(ctx, g):
        return ops.normalize(g.contiguous(), None, ctx.std), None, None


class _ReLUFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from . import _lib
        lib = _lib.load()
        y = torch.empty_like(x)
        _lib.check(lib.st_relu_fwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(y, "y"), x.numel(), _lib.stream_ptr()), "st_relu_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        (y,) = ctx.saved_tensors
        lib = _lib.load()
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.check(lib.st_relu_bwd(_lib.dev_ptr(y, "y"), _lib.dev_ptr(dy, "dy"), _lib.dev_ptr(dx, "dx"), y.numel(),
                                   _lib.stream_ptr()), "st_relu_bwd")
        return dx


class Normalize(nn.Module):
    """transforms.Normalize(mean, std) for batched NCHW tensors (trainer.py:299)."""

    def __init__(self, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        super().__init__()
        self.mean, self.std = tuple(mean), tuple(std)

    def forward(self, x):
        x = x.contiguous()
        if _needs_grad(x):
            return _NormalizeFunction.apply(x, self.mean, self.std)
        return ops.normalize(x, self.mean, self.std)


class VGG19FeatureExtractor(nn.Module):
    """``Trainer.get_feature_extractor(selected_indices)`` (trainer.py:165-199).

    ``forward(x)`` returns an OrderedDict in layer order, one tensor per requested index (keys are
    the index as a string); ``list(extractor(x).values())`` is what the reference's loop reads.
    The network is frozen (it is in no optimizer): only data gradients flow through it.
    """

    def __init__(self, selected_indices: Sequence[int], weights: Union[str, Dict[str, torch.Tensor], None] = None,
                 seed: int = 0):
        super().__init__()
        if not isinstance(selected_indices, list):
            raise ValueError(f"\n Selected indices should be provided as a list of layer indices,"
                             f" you pass in {selected_indices}")
        layout = vgg19_layers()
        if any(idx >= len(layout) or idx < 0 for idx in selected_indices):
            raise ValueError(f"\n Invalid layer index provided, the len of the model is {len(layout)}")
        self.selected = sorted(set(int(i) for i in selected_indices))
        self.last = max(self.selected)
        self.layout = layout[: self.last + 1]
        sd = self._load(weights)
        g = torch.Generator().manual_seed(seed)
        self.convs = nn.ModuleDict()
        for idx, (kind, cin, cout) in enumerate(self.layout):
            if kind != "conv":
                continue
            conv = nn.Conv2d(cin, cout, 3, padding=1)
            if sd is not None:
                v, b = sd[f"{idx}.weight"].float(), sd[f"{idx}.bias"].float()
            else:
                v = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cin * 9))
                b = torch.zeros(cout)
            # weight_norm(dim=0) with g initialised to ||v|| (trainer.py:179-181): w = g * v / ||v||
            norm = v.reshape(cout, -1).norm(dim=1).view(-1, 1, 1, 1)
            with torch.no_grad():
                conv.weight.copy_(v * (norm / norm))
                conv.bias.copy_(b)
            self.convs[str(idx)] = conv
        for p in self.parameters():
            p.requires_grad_(False)
        self._packed = _PackedWeightCache()

    @staticmethod
    def _load(weights):
        if weights is None:
            warnings.warn("VGG19FeatureExtractor: no ImageNet weights given (torchvision's IMAGENET1K_V1 is a "
                          "download); using a seeded random initialisation")
            return None
        sd = weights if isinstance(weights, dict) else torch.load(weights, map_location="cpu", weights_only=True)
        # accept torchvision's full-model keys ("features.0.weight") or bare ones ("0.weight")
        return {k[len("features."):] if k.startswith("features.") else k: v for k, v in sd.items()}

    def forward_multi(self, x, index_sets: Sequence[Sequence[int]]) -> List[List[torch.Tensor]]:
        """One pass that serves several extractors at once (the reference runs the content and the
        style extractor separately on the same decoded image, trainer.py:447,454).  Each set gets
        exactly what ITS extractor would return: a conv index that is the last layer of its own
        truncated model is captured pre-ReLU, every other captured conv post-ReLU."""
        sets = [sorted(set(int(i) for i in s)) for s in index_sets]
        last_all = max(max(s) for s in sets)
        assert last_all <= self.last, "extractor was built for a shallower network"
        need_pre = {max(s) for s in sets if self.layout[max(s)][0] == "conv"}
        compute = precision.get_compute_dtype()
        if compute in ("bf16", "bf16x3") and not _needs_grad(x):
            return self._forward_multi_g8(x, sets, last_all, need_pre, compute)
        if compute == "bf16" and self.GRAD_CHAIN and _needs_grad(x):
            return self._forward_multi_chain(x, sets, last_all)
        pre: Dict[int, torch.Tensor] = {}
        post: Dict[int, torch.Tensor] = {}
        idx = 0
        while idx <= last_all:
            kind = self.layout[idx][0]
            if kind == "conv":
                conv = self.convs[str(idx)]
                cont = idx + 1 <= last_all       # some extractor continues past this conv
                if idx in need_pre:
                    x = conv3x3_apply(x, conv.weight, conv.bias, self._packed, idx, upsample2x=False,
                                      pad_mode=ops.PAD_ZERO, relu=False)
                    pre[idx] = x
                    if cont:
                        x = _ReLUFunction.apply(x)
                        post[idx] = post[idx + 1] = x
                        idx += 1
                else:
                    x = conv3x3_apply(x, conv.weight, conv.bias, self._packed, idx, upsample2x=False,
                                      pad_mode=ops.PAD_ZERO, relu=True)
                    post[idx] = post[idx + 1] = x
                    idx += 1
            else:
                x = maxpool2x2_apply(x, ceil_mode=False)
                post[idx] = x
            idx += 1
        res = []
        for s in sets:
            res.append([pre[i] if (i == max(s) and i in pre) else post[i] for i in s])
        return res

    #: bf16 mode, differentiated pass (the decoded image): one autograd node whose forward AND backward hand
    #: activations / gradients over in G8 (autograd_vgg.VGGChainFunction); ST_VGG_CHAIN=0: the per-layer graph
    GRAD_CHAIN = os.environ.get("ST_VGG_CHAIN", "1") != "0"

    def _forward_multi_chain(self, x, sets, last_all):
        from .autograd_vgg import VGGChainFunction, capture_sets
        pre, post = capture_sets(self.layout, sets)
        keys = VGGChainFunction.keys(self.layout, last_all, pre, post)
        outs = dict(zip(keys, VGGChainFunction.apply(x, self, last_all, frozenset(pre), frozenset(post))))

        def key_of(s, i):
            if i == max(s) and self.layout[i][0] == "conv":
                return ("pre", i)
            # a post-ReLU capture belongs to the conv unit it sits in (conv index or its ReLU's index)
            return ("post", i - 1 if self.layout[i][0] == "relu" else i)
        # taps with exactly ONE consumer among the sets (and not a pool output) may receive their loss gradient as
        # per-plane affine coefficients (autograd_vgg.lazy_affine_grad; AdaINStyleLoss uses it)
        uses: Dict[tuple, int] = {}
        for s in sets:
            for i in s:
                uses[key_of(s, i)] = uses.get(key_of(s, i), 0) + 1
        for k, t in outs.items():
            if self.AFFINE_TAPS and uses.get(k, 0) == 1 and self.layout[k[1]][0] == "conv":
                t._st_chain_tap = True
        return [[outs[key_of(s, i)] for i in s] for s in sets]

    AFFINE_TAPS = os.environ.get("ST_NO_AFFINE_TAPS") is None

    def _forward_multi_g8(self, x, sets, last_all, need_pre, compute):
        """forward_multi without gradients in a bf16 compute mode (the passes over the content and style ORIGINALS,
        trainer.py:446-454): a conv whose output no extractor captures hands it to the next conv in the bf16
        channel-group layout G8 (ops.conv3x3_g8, LDS-DMA staged, zero padding patched after the copy), with the
        floor-mode 2x2 pool fused into its epilogue when the size is even (floor == ceil there); captured tensors
        are fp32 NCHW as always.  Same bits as the fp32-activation kernels."""
        from . import autograd_vgg
        want = {i for s in sets for i in s}
        pre: Dict[int, torch.Tensor] = {}
        post: Dict[int, torch.Tensor] = {}
        x = x.contiguous()
        idx = 0
        while idx <= last_all:
            kind, cin, cout = self.layout[idx]
            if kind != "conv":
                c_, h_, w_ = (int(v) for v in x.shape[1:])
                if (compute == "bf16" and idx not in want and idx + 1 <= last_all and h_ % 2 == 0 and w_ % 2 == 0
                        and c_ % 16 == 0):
                    x = ops.maxpool2x2_g8(x)               # uncaptured: straight into the next conv's input form
                else:
                    x = ops.maxpool2x2(x, False)
                    post[idx] = x
                idx += 1
                continue
            conv = self.convs[str(idx)]
            bias = conv.bias.detach()
            cont = idx + 1 <= last_all                     # the in-place ReLU after this conv is part of the pass
            if compute == "bf16x3" and 3 in (cin, cout):   # exact fp32 vector-ALU kernel (see _PackedWeightCache.get)
                x = ops.conv3x3(x, self._packed.get(idx, conv.weight, False, "f32"), bias, cout, False, ops.PAD_ZERO,
                                relu=cont and idx not in need_pre)
                if idx in need_pre:
                    pre[idx] = x
                    if cont:
                        x = torch.relu(x)
                if cont:
                    post[idx] = post[idx + 1] = x
                    idx += 1
                idx += 1
                continue
            wp = self._packed.get(idx, conv.weight, False, compute)
            if idx in need_pre:
                x = ops.conv3x3_g8(x, wp, bias, cout, False, ops.PAD_ZERO, relu=False, pool=False, y_g8=False)
                pre[idx] = x
                if cont:
                    x = torch.relu(x)
                    post[idx] = post[idx + 1] = x
                    idx += 1
                idx += 1
                continue
            captured = idx in want or idx + 1 in want
            h, w = int(x.shape[-3] if ops.is_g8(x) or ops.is_g8_pair(x) else x.shape[-2]), \
                int(x.shape[-2] if ops.is_g8(x) or ops.is_g8_pair(x) else x.shape[-1])
            pool_next = idx + 2 <= last_all and self.layout[idx + 2][0] == "pool"
            fuse_pool = pool_next and not captured and (idx + 2) not in want and h % 2 == 0 and w % 2 == 0
            if (autograd_vgg.TAP_POOL and compute == "bf16" and captured and pool_next and (idx + 2) not in want
                    and idx + 3 <= last_all and ops.is_g8(x) and h % 2 == 0 and w % 2 == 0 and cout % 16 == 0):
                # a tap that feeds a pool: the fp32 tap and the next conv's pooled G8 input from one launch
                y, x = ops.conv3x3_g8_tap_pool(x, wp, bias, cout, ops.PAD_ZERO, relu=True)
                post[idx] = post[idx + 1] = y
                idx += 3                                    # conv + ReLU + pool
                continue
            y_g8 = not captured and cout % 8 == 0 and (not pool_next or fuse_pool) and idx + 2 <= last_all
            x = ops.conv3x3_g8(x, wp, bias, cout, False, ops.PAD_ZERO, relu=True, pool=fuse_pool, y_g8=y_g8)
            if not y_g8:
                post[idx] = post[idx + 1] = x
            idx += 2                                        # conv + its ReLU
            if fuse_pool:
                idx += 1                                    # ... + the pool
        res = []
        for s in sets:
            res.append([pre[i] if (i == max(s) and i in pre) else post[i] for i in s])
        return res

    def forward(self, x) -> "OrderedDict[str, torch.Tensor]":
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        idx = 0
        n = len(self.layout)
        while idx < n:
            kind = self.layout[idx][0]
            if kind == "conv":
                fuse_relu = idx + 1 < n          # the in-place ReLU that follows exists in the truncated model
                conv = self.convs[str(idx)]
                x = conv3x3_apply(x, conv.weight, conv.bias, self._packed, idx, upsample2x=False,
                                  pad_mode=ops.PAD_ZERO, relu=fuse_relu)
                if idx in self.selected:
                    out[str(idx)] = x            # post-ReLU unless this conv is the last layer
                if fuse_relu:
                    if idx + 1 in self.selected:
                        out[str(idx + 1)] = x
                    idx += 1
            elif kind == "pool":
                x = maxpool2x2_apply(x, ceil_mode=False)
                if idx in self.selected:
                    out[str(idx)] = x
            else:  # a ReLU not preceded by a conv cannot occur in VGG19
                raise RuntimeError("unexpected layer order")
            idx += 1
        return OrderedDict((str(i), out[str(i)]) for i in self.selected)


def get_feature_extractor(selected_indices, vgg_model_type: str = "19", device="cpu", weights=None):
    """Signature of Trainer.get_feature_extractor (trainer.py:166); only "19" is ever used there."""
    if vgg_model_type != "19":
        raise ValueError("only the VGG19 loss network of the reference's build_model is implemented")
    return VGG19FeatureExtractor(selected_indices, weights=weights).to(device)
