"""ctypes binding of libst_hip.so (C ABI declared in include/st_hip.h).

The product path has no CPU fallback: if the library is missing or a tensor is
not a contiguous fp32 HIP tensor, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_size_t, c_void_p

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libst_hip.so")

# name -> (restype, argtypes); every symbol include/st_hip.h declares
SIGNATURES = {
    "st_version": (c_int, []),
    "st_last_error": (ctypes.c_char_p, []),
    "st_conv3x3_packed_elems": (c_size_t, [c_int, c_int]),
    "st_conv3x3_pack_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_up2_packed_elems": (c_size_t, [c_int, c_int]),
    "st_conv3x3_pack_weights_up2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "st_conv3x3_up2_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p]),
    "st_conv3x3_pool_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p]),
    "st_conv3x3_pool_fwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_int, c_void_p]),
    "st_conv3x3_fwd_bf16_g8": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_fwd_bf16x3_g8": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_dgrad_bf16_g8": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                         c_int, c_void_p]),
    "st_conv3x3_dgrad_bf16_g8_chain": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_dgrad_fold8_bias_slabs": (c_int, [c_int, c_int, c_int]),
    "st_conv3x3_dgrad_chain_bias_slabs": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "st_conv3x3_wgrad_bf16_pre_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "st_conv3x3_wgrad_bf16_pre": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                          c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "st_bmm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "st_conv3x3_wgrad_cout3_xg8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                           c_int, c_int, c_void_p]),
    "st_grad_join_g8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_void_p]),
    "st_grad_join_g8_affine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "st_meanstd_loss_bwd_coeffs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int,
                                           c_int, c_void_p]),
    "st_conv3x3_dgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "st_conv3x3_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_packed_elems_bf16": (c_size_t, [c_int, c_int]),
    "st_conv3x3_pack_weights_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_fwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_dgrad_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_packed_elems_bf16x3": (c_size_t, [c_int, c_int]),
    "st_conv3x3_pack_weights_bf16x3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_fwd_bf16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_pool_fwd_bf16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_int, c_void_p]),
    "st_conv3x3_dgrad_bf16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                        c_int, c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
"st_conv3x3_wgrad_bf16_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "st_conv3x3_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                 c_int, c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_wgrad_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_void_p]),
    "st_maxpool2x2_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "st_sympad_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "st_normalize_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "st_relu_fwd": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p]),
    "st_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p]),
    "st_conv1x1_3to3_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "st_maxpool2x2_fwd_g8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_pool_fwd_bf16_g8_codes": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                  c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "st_conv3x3_fwd_bf16_g8_tap_pool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                c_int, c_int, c_int, c_int, c_void_p]),
    "st_grad_pool_codes_g8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "st_maxpool2x2_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "st_adain_fwd_dev_eps": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float,
                                     c_void_p, c_void_p]),
    "st_adain_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float,
                             c_float, c_void_p, c_void_p]),
    "st_adain_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_int, c_int, c_int, c_int, c_void_p]),
    "st_reduce_workspace_bytes": (c_size_t, []),
    "st_mse_fwd": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_void_p]),
    "st_mse_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_longlong, c_int, c_void_p]),
    "st_tv_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "st_tv_bwd": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "st_meanstd_loss_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "st_meanstd_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                    c_int, c_int, c_int, c_void_p]),
    "st_rgbuv_hist_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "st_rgbuv_hist_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_float, c_void_p, c_int, c_int, c_void_p]),
    "st_rgbuv_hist_static_scratch_bytes": (c_size_t, []),
    "st_rgbuv_hist_fwd_static": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_float, c_void_p, c_int, c_int, c_void_p]),
    "st_rgbuv_hist_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_int, c_int,
                                  c_void_p]),
    "st_hellinger_sq_fwd": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_void_p]),
    "st_hellinger_sq_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_longlong, c_void_p]),
    "st_cov_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "st_cov_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "st_cov_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "st_meanstd_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "st_aug_gray_stack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "st_aug_rotate": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "st_aug_gaussian_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "st_aug_sharpness": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "st_aug_color_jitter_slot": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "st_aug_hflip": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "st_syevj_small_max_dim": (c_int, []),
    "st_syevj_small_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
}

_lib = None


class StHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """dlopen libst_hip.so and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StHipError(
            f"{LIB_PATH} not found: build it with `python -m style_transfer_amd.build` "
            "(there is no CPU fallback for the HIP hot path)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().st_last_error()
        raise StHipError(f"{what} failed ({code}): {msg.decode() if msg else ''}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device.  The raw accessors cost ~0.3 us; building a
    torch.cuda.Stream object per launch cost ~9 us, i.e. ~2 ms of host time per training step."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def dev_ptr(t: torch.Tensor, name: str = "tensor", dtype=torch.float32) -> int:
    """Raw device pointer of a contiguous HIP tensor of `dtype` (fp32 by default; loud failure otherwise)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise StHipError(f"{name}: the HIP hot path needs a tensor on the GPU, got device={t.device} "
                         "(no CPU fallback; the CPU oracle lives in oracle/ and is test-only)")
    if t.dtype != dtype:
        raise StHipError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise StHipError(f"{name}: expected a contiguous tensor")
    return t.data_ptr()
