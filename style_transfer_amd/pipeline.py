"""Composed hot paths: the inference composition of Trainer.plot_comparison
(src/models/trainer.py:695-708) as one call on batched tensors."""
from __future__ import annotations

import torch

from .generator import Decoder, Encoder
from .transformer import AdaIN

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


@torch.no_grad()
def stylize(encoder: Encoder, decoder: Decoder, transformer: AdaIN, content: torch.Tensor, style: torch.Tensor,
            alpha: float = 1.0) -> torch.Tensor:
    """encode C, encode S, AdaIN, alpha blend (fused), decode (trainer.py:698-704)."""
    fc = encoder(content)
    fs = encoder(style)
    t = transformer(fc, fs, alpha=alpha)
    return decoder(t)
