"""Composed hot paths: the inference composition of Trainer.plot_comparison
(src/models/trainer.py:695-708) as one call on batched tensors."""
from __future__ import annotations

import torch

from .generator import Decoder, Encoder
from .transformer import AdaIN

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
BATCH_ENCODER_BELOW = 8


@torch.no_grad()
def stylize(encoder: Encoder, decoder: Decoder, transformer: AdaIN, content: torch.Tensor, style: torch.Tensor,
            alpha: float = 1.0) -> torch.Tensor:
    """encode C, encode S, AdaIN, alpha blend (fused), decode (trainer.py:698-704).

    Content and style of equal shape go through the frozen encoder as ONE batch (as the training step does,
    trainer.py:211-216 of this package): per-image results are identical (no cross-image op in the encoder) and a
    small per-rank batch fills twice the grid.  ``BATCH_ENCODER_BELOW`` images per side: from there on each pass
    fills the chip on its own and the concatenation copy would only add traffic."""
    if content.shape == style.shape and content.shape[0] < BATCH_ENCODER_BELOW:
        fc, fs = encoder(torch.cat((content, style), dim=0)).split(content.shape[0], dim=0)
    else:
        fc = encoder(content)
        fs = encoder(style)
    t = transformer(fc, fs, alpha=alpha)
    return decoder(t)
