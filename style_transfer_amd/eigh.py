"""Batched symmetric eigendecomposition for VincentStyleLoss.l2wass_dist (src/models/losses.py:67-81), as a direct
binding of rocSOLVER's ``rocsolver_ssyevd_strided_batched`` -- the same driver ``torch.linalg.eigh`` reaches on ROCm
(bit-identical eigenvalues), minus what torch wraps around it:

* ``torch.linalg.eigh`` reads the solver's ``info`` back on the host after EVERY call.  That host synchronisation is
  what kept the five per-layer decompositions of a training step strictly serial (measured: 34.5 ms back to back, and
  still 34.9 ms when issued on five streams through torch, 23.2 ms on five streams through this binding).  Here
  ``info`` stays on the device; ``check()`` reads every pending one at once, when the caller chooses (the trainer:
  once per epoch).
* the decomposition is latency-bound (thousands of tiny launches): [16, 512, 512] costs 13.0 ms against 11.5 ms for
  [8, 512, 512], so callers batch the layers that share a channel count.

SURVEY.md K7b: the eigensolver itself stays rocSOLVER (LAPACK-class work); the hand-written part of the Gram loss is
the covariance (csrc/cov.hip).  One rocBLAS handle per HIP stream: a handle owns the solver's device workspace, so
calls that may overlap must not share one.
"""
from __future__ import annotations

import ctypes
from ctypes import c_int, c_longlong, c_void_p
from typing import Dict, List, Tuple

import os

import torch

from . import _lib

_EVECT_ORIGINAL, _FILL_LOWER = 211, 122       # rocblas_evect_original, rocblas_fill_lower
_rs = _rb = None
_handles: Dict[Tuple[int, int], c_void_p] = {}
_pending: List[torch.Tensor] = []
_FOLD_AT = 1024     # pending status tensors are folded into ONE device-side maximum at this count (nothing is dropped)


def _keep(pending: List[torch.Tensor], status: torch.Tensor) -> None:
    """Queue a per-call status tensor for check().  A long epoch (steps x groups) must neither grow the list without
    bound nor lose entries: at _FOLD_AT entries the list collapses into one 1-element maximum, on the device, with no
    host synchronisation (status values are >= 0 in both solvers' conventions after abs())."""
    pending.append(status)
    if len(pending) >= _FOLD_AT and not torch.cuda.is_current_stream_capturing():
        by_dev: Dict[torch.device, List[torch.Tensor]] = {}
        for t in pending:
            by_dev.setdefault(t.device, []).append(t.abs().max().reshape(1))
        pending[:] = [torch.cat(v).max().reshape(1) for v in by_dev.values()]


def _load():
    global _rs, _rb
    if _rs is None:
        _rb = ctypes.CDLL("librocblas.so")
        _rs = ctypes.CDLL("librocsolver.so")
        _rb.rocblas_create_handle.argtypes = [ctypes.POINTER(c_void_p)]
        _rb.rocblas_set_stream.argtypes = [c_void_p, c_void_p]
        _rs.rocsolver_ssyevd_strided_batched.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_longlong,
                                                         c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_int]
    return _rs


def _handle(stream: int) -> c_void_p:
    key = (torch.cuda.current_device(), stream)
    h = _handles.get(key)
    if h is None:
        h = c_void_p()
        if _rb.rocblas_create_handle(ctypes.byref(h)) != 0:
            raise _lib.StHipError("rocblas_create_handle failed")
        if _rb.rocblas_set_stream(h, c_void_p(stream)) != 0:
            raise _lib.StHipError("rocblas_set_stream failed")
        _handles[key] = h
    return h


MAX_SWEEPS = 20
# Largest C routed to the hand-written Jacobi kernel.  Measured at [8, C, C]: C = 64 beats rocSOLVER's syevd several
# times over; at C = 128 one CU's LDS bandwidth bounds a Jacobi round (2 x 128 x 128 element updates, 127 rounds
# per sweep, 6 sweeps) at more than rocSOLVER's 1.9 ms, so 128 and up stay on rocSOLVER.
SMALL_MAX = 64
_small_pending: List[torch.Tensor] = []


def syevj_small(a: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """The same through the hand-written LDS-resident Jacobi kernel (csrc/eigh.hip; any even C <= 128, routed here by
    ``syevd`` for C <= SMALL_MAX = 64), one workgroup per matrix, ONE launch instead of rocSOLVER's hundreds."""
    lib = _lib.load()
    b, c, _ = (int(v) for v in a.shape)
    a = a.detach().contiguous()
    w = torch.empty((b, c), dtype=torch.float32, device=a.device)
    v = torch.empty((b, c, c), dtype=torch.float32, device=a.device)
    sweeps = torch.empty(b, dtype=torch.int32, device=a.device)
    _lib.check(lib.st_syevj_small_batched(_lib.dev_ptr(a, "a"), _lib.dev_ptr(w, "w"), _lib.dev_ptr(v, "v"),
                                          _lib.dev_ptr(sweeps, "sweeps", torch.int32), b, c, MAX_SWEEPS,
                                          _lib.stream_ptr()), "st_syevj_small_batched")
    _keep(_small_pending, sweeps)
    return w, v


def syevd(a: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eigenvalues (ascending) [B, C] and eigenvectors V [B, C, C] (columns) of the symmetric matrices ``a``
    [B, C, C], on torch's current stream, without any host synchronisation.  Even C <= SMALL_MAX (64):
    ``syevj_small``; larger: rocSOLVER's syevd."""
    assert a.dim() == 3 and a.shape[1] == a.shape[2], f"expected [B, C, C], got {tuple(a.shape)}"
    b, c, _ = (int(v) for v in a.shape)
    if c % 2 == 0 and 2 <= c <= SMALL_MAX:
        return syevj_small(a)
    rs = _load()
    work = a.detach().clone(memory_format=torch.contiguous_format)     # overwritten with the eigenvectors
    w = torch.empty((b, c), dtype=torch.float32, device=a.device)
    e = torch.empty((b, c), dtype=torch.float32, device=a.device)
    info = torch.zeros(b, dtype=torch.int32, device=a.device)
    rc = rs.rocsolver_ssyevd_strided_batched(_handle(_lib.stream_ptr()), _EVECT_ORIGINAL, _FILL_LOWER, c,
                                             _lib.dev_ptr(work, "a"), c, c * c, _lib.dev_ptr(w, "w"), c,
                                             _lib.dev_ptr(e, "e"), c, _lib.dev_ptr(info, "info", torch.int32), b)
    if rc != 0:
        raise _lib.StHipError(f"rocsolver_ssyevd_strided_batched returned status {rc}")
    _keep(_pending, info)
    # column-major output: the buffer read row-major holds V^T (rows = eigenvectors)
    return w, work.transpose(-1, -2)


def check() -> None:
    """Read every ``info`` issued since the last check (ONE host synchronisation) and raise if a decomposition did
    not converge -- what torch.linalg.eigh does after each call."""
    global _pending, _small_pending
    if _small_pending:
        sw, _small_pending = _small_pending, []
        # the kernel reports the sweeps it used, or MAX_SWEEPS + 1 when it ran out without a rotation-free sweep
        worst = max(int(s.max()) for s in sw)
        if worst > MAX_SWEEPS:
            raise _lib.StHipError(f"linalg.eigh (Jacobi, C <= {SMALL_MAX}): no convergence within {MAX_SWEEPS} sweeps")
    if not _pending:
        return
    infos, _pending = _pending, []
    bad = max(int(i.abs().max()) for i in infos)
    if bad != 0:
        raise _lib.StHipError(f"linalg.eigh: the algorithm failed to converge ({bad} off-diagonal elements of an "
                              "intermediate tridiagonal form did not converge to zero)")


HIP_BMM = os.environ.get("ST_NO_HIP_BMM") is None      # V diag(g) V^T on st_bmm_f32 (csrc/bmm.hip)


class _EigvalshFunction(torch.autograd.Function):
    """torch.linalg.eigvalsh with its backward (d lambda_i / dA = v_i v_i^T): gA = V diag(g) V^T."""

    @staticmethod
    def forward(ctx, a):
        w, v = syevd(a)
        ctx.save_for_backward(v)
        return w

    @staticmethod
    def backward(ctx, g):
        (v,) = ctx.saved_tensors
        scaled = v * g.unsqueeze(-2)
        from . import ops
        if HIP_BMM and ops.bmm_ok(scaled):
            return ops.bmm_raw(scaled, v, False, True)
        return scaled @ v.transpose(-1, -2)


def eigvalsh(a: torch.Tensor) -> torch.Tensor:
    return _EigvalshFunction.apply(a)
