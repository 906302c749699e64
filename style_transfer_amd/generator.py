"""Encoder / Decoder / SymmetricPadding2D with the reference's names, constructor
signatures, forward signatures and state_dict keys (src/models/generator.py:14-130),
running on hand-written HIP kernels (libst_hip.so) instead of stock ATen dispatch.

The ``nn.Sequential`` containers keep the reference's module positions so
``state_dict()`` keys (``encode.{0,2,...,29}.*`` / ``decoder.{1,...,28}.*``) and
``.pth`` files are interchangeable, but ``forward`` never runs the Sequential: each
[SymmetricPadding2D -> Conv2d(3x3) -> ReLU] triple (and a preceding nearest x2
upsample) is ONE fused MFMA implicit-GEMM launch.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import ops, precision
from .autograd import _needs_grad, conv3x3_apply, maxpool2x2_apply, conv1x1_apply, sympad_apply

ENCODER_WEIGHTS = r'./src/models/checkpoints/WCT_encoder_decoder/vgg_normalised_conv4_1.pth'   # generator.py:78
DECODER_WEIGHTS = r"./src/models/checkpoints/WCT_encoder_decoder/feature_invertor_conv4_1.pth"  # generator.py:121


class SymmetricPadding2D(nn.Module):
    """generator.py:14-39.  padding = (left, right, top, bottom)."""

    def __init__(self, padding: Tuple[int, int, int, int] = (1, 1, 1, 1)):
        super().__init__()
        self.padding = padding

    def forward(self, im: torch.Tensor):
        return sympad_apply(im, self.padding)


class _PackedWeightCache:
    """Packed (K-major) copies of OIHW conv weights; rebuilt when the parameter changes.

    Freshness = (parameter ``_version``, its ``data_ptr``, the process-wide ``generation``).  A HIP-graph replay
    moves weights in place through captured optimizer kernels WITHOUT bumping ``_version``; GraphedStep therefore
    calls ``invalidate_all()`` after every replay, so an eager forward that follows (per-epoch eval, a ragged last
    batch, ``Trainer.stylize`` after ``train()``) repacks instead of reusing a stale copy."""

    generation = 0

    @classmethod
    def invalidate_all(cls) -> None:
        cls.generation += 1

    def __init__(self):
        self._c: Dict[Tuple[int, bool], Tuple[int, int, torch.Tensor]] = {}
        self.direct_grads = False     # set by the trainer: weight gradients go straight into its flat bucket

    #: fp32 mode: a layer that reads its input through the x2 upsample runs its FORWARD in the sub-pixel form (four
    #: 2 x 2 convolutions over the low-resolution input, st_conv3x3_up2_fwd: 16 / 36 of the multiply-adds);
    #: ST_NO_UP2=1: the direct 3 x 3 form on the upsampled input
    UP2 = os.environ.get("ST_NO_UP2") is None

    def get(self, key, w: torch.Tensor, transpose_flip: bool = False, compute: Optional[str] = None,
            forward_only: bool = False, upsample2x: bool = False) -> torch.Tensor:
        """``upsample2x``: the forward weights of a replicate-padded layer behind the x2 upsample (fp32 mode: the
        sub-pixel packing, see UP2)."""
        compute = compute or precision.get_compute_dtype()
        if upsample2x and compute == "f32" and not transpose_flip and self.UP2:
            compute = "f32up2"
        if compute == "bf16x3" and forward_only and 3 in (int(w.shape[0]), int(w.shape[1])):
            # the 3-channel layers are HBM-bound: the exact-fp32 vector-ALU kernels (csrc/conv3x3_small.hip)
            # are as fast as any matrix-pipe form, so the fp32-grade mode keeps them exact (forward-only callers:
            # a layer that will be differentiated stays on the split kernels so that its backward does too)
            compute = "f32"
        ent = self._c.get((key, transpose_flip, compute))
        fresh = (w._version, w.data_ptr(), _PackedWeightCache.generation)
        if ent is not None and ent[0] == fresh:
            return ent[1]
        packed = ops.pack_conv3x3_weights(w, transpose_flip, compute)
        self._c[(key, transpose_flip, compute)] = (fresh, packed)
        return packed

    def clear(self):
        self._c.clear()


def _g8_chain(x, seq, idxs, packed, compute, pool_after=(), up_after=(), relu_last=True):
    """Forward-only [pad, conv3x3, ReLU] chain in a bf16 compute mode: the convs hand their activations to each
    other in the bf16 channel-group layout G8 (bf16x3: the hi/lo pair form) -- ops.conv3x3_g8: LDS-DMA staging, half
    the HBM bytes in bf16 mode, bit-identical to the fp32-activation kernels.  fp32 NCHW at both ends.  In bf16x3
    mode the 3-channel layers stay on the exact fp32 vector-ALU kernels (see _PackedWeightCache.get)."""
    def small(idx):
        return compute == "bf16x3" and 3 in (int(seq[idx].weight.shape[0]), int(seq[idx].weight.shape[1]))

    up = False
    for k, idx in enumerate(idxs):
        conv = seq[idx]
        cout = int(conv.weight.shape[0])
        last = k == len(idxs) - 1
        relu = relu_last or not last
        if small(idx):
            assert idx not in pool_after
            x = ops.conv3x3(x, packed.get(idx, conv.weight, False, "f32"), conv.bias.detach(), cout, up,
                            ops.PAD_REPLICATE, relu)
        else:
            x = ops.conv3x3_g8(x, packed.get(idx, conv.weight, False, compute), conv.bias.detach(), cout,
                               upsample2x=up, pad_mode=ops.PAD_REPLICATE, relu=relu, pool=idx in pool_after,
                               y_g8=not last and not small(idxs[k + 1]))
        up = idx in up_after     # the next conv reads this output through a x2 nearest upsample
    return x


def _load_state(path_or_sd: Union[str, dict]) -> dict:
    if isinstance(path_or_sd, dict):
        return path_or_sd
    return torch.load(path_or_sd, map_location="cpu", weights_only=True)


class Encoder(nn.Module):
    """VGG19-normalised encoder up to relu4_1 (generator.py:42-82).

    ``Encoder()`` behaves like the reference: it loads the cwd-relative
    ``vgg_normalised_conv4_1.pth`` and raises FileNotFoundError when it is missing.
    ``weights`` may instead be another path, a state dict with the bare Sequential keys
    (``"0.weight"`` ...), or ``False`` to keep the random initialisation.
    The encoder is frozen (no optimizer in the reference ever reads its gradients,
    trainer.py:286-297), so its parameters do not require grad.
    """

    POOL_AFTER = (5, 12, 25)   # ceil-mode pools at Sequential indices 7, 14, 27

    def __init__(self, weights: Union[str, dict, bool, None] = None):
        super().__init__()
        self.encode = nn.Sequential(
            nn.Conv2d(3, 3, (1, 1)),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(3, 64, (3, 3)), nn.ReLU(),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(64, 64, (3, 3)), nn.ReLU(),
            nn.MaxPool2d((2, 2), (2, 2), (0, 0), ceil_mode=True),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(64, 128, (3, 3)), nn.ReLU(),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(128, 128, (3, 3)), nn.ReLU(),
            nn.MaxPool2d((2, 2), (2, 2), (0, 0), ceil_mode=True),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(128, 256, (3, 3)), nn.ReLU(),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(256, 256, (3, 3)), nn.ReLU(),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(256, 256, (3, 3)), nn.ReLU(),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(256, 256, (3, 3)), nn.ReLU(),
            nn.MaxPool2d((2, 2), (2, 2), (0, 0), ceil_mode=True),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(256, 512, (3, 3)), nn.ReLU(),
        )
        if weights is not False:
            self.encode.load_state_dict(_load_state(ENCODER_WEIGHTS if weights is None else weights))
        for p in self.parameters():
            p.requires_grad_(False)
        self._packed = _PackedWeightCache()

    CONV3 = (2, 5, 9, 12, 16, 19, 22, 25, 29)

    def forward(self, x):
        seq = self.encode
        x = conv1x1_apply(x, seq[0].weight, seq[0].bias)
        frozen = not (torch.is_grad_enabled() and x.requires_grad)
        if frozen and precision.get_compute_dtype() in ("bf16", "bf16x3"):
            return _g8_chain(x, seq, self.CONV3, self._packed, precision.get_compute_dtype(),
                             pool_after=self.POOL_AFTER)
        for idx in self.CONV3:
            conv = seq[idx]
            if idx in self.POOL_AFTER and frozen:
                # [pad, conv, ReLU, MaxPool(ceil)] in one launch: the full-resolution activation is never written
                x = ops.conv3x3_pool(x, self._packed.get(idx, conv.weight, False), conv.bias.detach(),
                                     int(conv.weight.shape[0]), pad_mode=ops.PAD_REPLICATE, relu=True)
                continue
            x = conv3x3_apply(x, conv.weight, conv.bias, self._packed, idx, upsample2x=False,
                              pad_mode=ops.PAD_REPLICATE, relu=True)
            if idx in self.POOL_AFTER:
                x = maxpool2x2_apply(x, ceil_mode=True)
        return x


class Decoder(nn.Module):
    """Mirrored decoder (generator.py:85-130): keys ``decoder.{1,5,8,11,14,18,21,25,28}``."""

    CONV3 = (1, 5, 8, 11, 14, 18, 21, 25, 28)
    UP_AFTER = (1, 14, 21)
    #: bf16 mode, training: G8 hand-overs in both directions (autograd_conv.DecoderChainG8Function); ST_DEC_CHAIN=0:
    #: the fp32-activation chain
    G8_CHAIN = os.environ.get("ST_DEC_CHAIN", "1") != "0"

    def __init__(self, use_pretrained_WCT: bool = False, do_train: bool = False,
                 weights: Union[str, dict, None] = None):
        super().__init__()
        self.decoder = nn.Sequential(
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(512, 256, (3, 3)), nn.ReLU(),
            nn.UpsamplingNearest2d(scale_factor=2),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(256, 256, (3, 3)), nn.ReLU(),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(256, 256, (3, 3)), nn.ReLU(),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(256, 256, (3, 3)), nn.ReLU(),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(256, 128, (3, 3)), nn.ReLU(),
            nn.UpsamplingNearest2d(scale_factor=2),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(128, 128, (3, 3)), nn.ReLU(),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(128, 64, (3, 3)), nn.ReLU(),
            nn.UpsamplingNearest2d(scale_factor=2),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(64, 64, (3, 3)), nn.ReLU(),
            SymmetricPadding2D((1, 1, 1, 1)), nn.Conv2d(64, 3, (3, 3)),
        )
        if use_pretrained_WCT:
            print("\n Using WCT pretrained image recover...")
            self.decoder.load_state_dict(_load_state(DECODER_WEIGHTS if weights is None else weights))
        if do_train:
            print("\n Enabling Decoder training...")
            self.decoder.train()
        else:
            self.decoder.eval()
        self._packed = _PackedWeightCache()

    def forward(self, x):
        seq = self.decoder
        params = [t for idx in self.CONV3 for t in (seq[idx].weight, seq[idx].bias)]
        if _needs_grad(x, *params):
            # training: the whole stack is one autograd node (fused ReLU backward, autograd_conv.py)
            from .autograd_conv import DecoderChainFunction
            specs, up = [], False
            for idx in self.CONV3:
                specs.append((idx, int(seq[idx].weight.shape[0]), up, idx != 28))
                up = idx in self.UP_AFTER
            assert not specs[-1][3], "the chain backward relies on the last conv having no ReLU"
            if precision.get_compute_dtype() == "bf16" and self.G8_CHAIN:
                from .autograd_conv import DecoderChainG8Function
                if DecoderChainG8Function.usable(x, specs, params):
                    return DecoderChainG8Function.apply(x.contiguous(), self._packed, tuple(specs), *params)
            return DecoderChainFunction.apply(x.contiguous(), self._packed, tuple(specs), *params)
        up = False
        if precision.get_compute_dtype() in ("bf16", "bf16x3"):
            return _g8_chain(x.contiguous(), seq, self.CONV3, self._packed, precision.get_compute_dtype(),
                             up_after=self.UP_AFTER, relu_last=False)
        for idx in self.CONV3:
            conv = seq[idx]
            x = conv3x3_apply(x, conv.weight, conv.bias, self._packed, idx, upsample2x=up,
                              pad_mode=ops.PAD_REPLICATE, relu=(idx != 28))
            up = idx in self.UP_AFTER     # the next conv reads this output through a x2 nearest upsample
        return x
