"""Compute precision of the convolution stacks.

``"f32"`` (default): exact fp32 MFMA (v_mfma_f32_32x32x2_f32), the reference's numerics.
``"bf16"``: bf16 MFMA operands with fp32 accumulation (v_mfma_f32_32x32x16_bf16), fp32 master
weights and fp32 NCHW activations on every module boundary (BASELINE config 4).  Only the 3x3
convolutions (forward and both gradients) change; statistics, losses and reductions stay fp32.
``"bf16x3"``: fp32-grade results from the bf16 matrix pipe: every fp32 operand is split exactly into
hi = bf16(v), lo = bf16(v - hi) and each product is hi*hi + hi*lo + lo*hi with fp32 accumulation
(three bf16 MFMAs per product, still ~5x the fp32 MFMA rate; ~1e-5 relative error on the whole
stylise pipeline against the 1e-3 fp32 tolerance).  Forward and data gradient use it; the weight
gradient stays on the exact fp32 kernel.

    from style_transfer_amd import precision
    precision.set_compute_dtype("bf16")          # process-wide
    with precision.compute_dtype("bf16"): ...    # scoped
"""
from __future__ import annotations

import contextlib

_VALID = ("f32", "bf16", "bf16x3")
_compute = "f32"


def set_compute_dtype(name: str) -> None:
    global _compute
    if name not in _VALID:
        raise ValueError(f"compute dtype must be one of {_VALID}, got {name!r}")
    _compute = name


def get_compute_dtype() -> str:
    return _compute


@contextlib.contextmanager
def compute_dtype(name: str):
    prev = get_compute_dtype()
    set_compute_dtype(name)
    try:
        yield
    finally:
        set_compute_dtype(prev)
