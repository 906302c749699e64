"""Compute precision of the convolution stacks.

``"f32"`` (default): exact fp32 MFMA (v_mfma_f32_32x32x2_f32), the reference's numerics.
``"bf16"``: bf16 MFMA operands with fp32 accumulation (v_mfma_f32_32x32x16_bf16), fp32 master
weights and fp32 NCHW activations on every module boundary (BASELINE config 4).  Only the 3x3
convolutions (forward and data gradient) change; statistics, losses and reductions stay fp32.

    from style_transfer_amd import precision
    precision.set_compute_dtype("bf16")          # process-wide
    with precision.compute_dtype("bf16"): ...    # scoped
"""
from __future__ import annotations

import contextlib

_VALID = ("f32", "bf16")
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
