"""Functional host wrappers over the C ABI (include/st_hip.h): shape checks, output
allocation through torch's caching allocator, raw pointers + current HIP stream.

Forward-only entry points live here; autograd wiring is in ``autograd.py``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

PAD_REPLICATE, PAD_ZERO = 0, 1


class _KernelTimer:
    """Optional per-launch HIP-event timing (used by bench.py for the roofline object).
    Events are recorded on torch's current stream, the stream every kernel is launched on."""

    def __init__(self):
        self.enabled = False
        self._pending = []
        self._pool = []

    def start(self, prealloc: int = 0):
        """``prealloc``: number of launches expected; their events are created here, outside any
        timed region (creating HIP events inside it showed up as launch gaps on a fresh box)."""
        self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * prealloc)]
        self.enabled = True
        self._pending = []

    def _event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def record(self, kernel: str, flops: float, nbytes: float):
        e0 = self._event()
        e1 = self._event()
        self._pending.append((kernel, flops, nbytes, e0, e1))
        e0.record()
        return e1

    def stop(self):
        self.enabled = False
        torch.cuda.synchronize()
        out = [{"kernel": k, "flops": f, "bytes": b, "ms": e0.elapsed_time(e1)} for k, f, b, e0, e1 in self._pending]
        self._pending = []
        return out


KERNEL_TIMER = _KernelTimer()


def conv_family(w_packed: torch.Tensor, cout: int, cin: int) -> str:
    """Kernel family a packed weight buffer belongs to: "f32" (fp32 buffer), "bf16" or "bf16x3" (bf16
    buffers; the x3 one holds the hi and the lo image, twice the elements).  Checks the size."""
    lib = _lib.load()
    if w_packed.dtype == torch.float32:
        fam, need = "f32", lib.st_conv3x3_packed_elems(cout, cin)
        if w_packed.numel() == lib.st_conv3x3_up2_packed_elems(cout, cin):
            fam, need = "f32up2", w_packed.numel()       # sub-pixel packing of an upsample layer (forward only)
    else:
        assert w_packed.dtype == torch.bfloat16, f"packed weights must be fp32 or bf16, got {w_packed.dtype}"
        need = lib.st_conv3x3_packed_elems_bf16(cout, cin)
        fam = "bf16"
        if w_packed.numel() == 2 * need:
            fam, need = "bf16x3", 2 * need
    assert w_packed.numel() == need, f"packed weights have {w_packed.numel()} elements, expected {need}"
    return fam


_SUFFIX = {"f32": "", "f32up2": "", "bf16": "_bf16", "bf16x3": "_bf16x3"}


def conv_kernel_name(cout: int, bf16=False) -> str:
    """Template instantiation st_conv3x3_fwd[_bf16[x3]] dispatches to (csrc/conv3x3.hip, conv3x3_bf16.hip);
    ``bf16``: bool or a family name."""
    if bf16 == "bf16x3":
        return "conv3x3_bf16x3_kernel<1,4,2,4>"
    if bf16 == "f32up2":
        return "conv3x3_f32_kernel<2,2,sub-pixel>"
    if bf16 == "f32":
        bf16 = False
    if bf16:
        return "conv3x3_bf16_kernel<2,4,2,4>" if cout > 64 else (
            "conv3x3_bf16_kernel<2,4,1,8>" if cout > 32 else "conv3x3_bf16_kernel<1,4,1,8>")
    if cout > 64:
        return "conv3x3_f32_kernel<2,2,2,2>"
    if cout > 32:
        return "conv3x3_f32_kernel<2,2,1,4>"
    return "conv3x3_f32_kernel<1,4,1,4>"
_PAD = {"replicate": PAD_REPLICATE, "symmetric": PAD_REPLICATE, "zero": PAD_ZERO, "zeros": PAD_ZERO}


def _pad_code(pad_mode) -> int:
    if isinstance(pad_mode, int):
        return pad_mode
    return _PAD[pad_mode]


def pack_conv3x3_weights(w: torch.Tensor, transpose_flip: bool = False, compute: str = "f32") -> torch.Tensor:
    """OIHW [cout, cin, 3, 3] -> the packed layout st_conv3x3_fwd consumes: K-major fp32 for
    ``compute="f32"``, channel-innermost bf16 for ``compute="bf16"`` (the dtype of the returned
    tensor selects the kernel family in conv3x3 / conv3x3_dgrad), hi + lo bf16 images for "bf16x3"."""
    lib = _lib.load()
    assert w.dim() == 4 and w.shape[2] == 3 and w.shape[3] == 3, f"expected OIHW 3x3 weights, got {tuple(w.shape)}"
    cout, cin = int(w.shape[0]), int(w.shape[1])
    w = w.detach()
    pc_out, pc_in = (cin, cout) if transpose_flip else (cout, cin)
    if compute == "f32up2":
        assert not transpose_flip, "the sub-pixel packing is a forward-only form"
        packed = torch.empty(lib.st_conv3x3_up2_packed_elems(cout, cin), dtype=torch.float32, device=w.device)
        _lib.check(lib.st_conv3x3_pack_weights_up2(_lib.dev_ptr(w, "w"), _lib.dev_ptr(packed, "packed"), cout, cin,
                                                   _lib.stream_ptr()), "st_conv3x3_pack_weights_up2")
        return packed
    if compute in ("bf16", "bf16x3"):
        elems = getattr(lib, "st_conv3x3_packed_elems" + _SUFFIX[compute])
        pack = getattr(lib, "st_conv3x3_pack_weights" + _SUFFIX[compute])
        packed = torch.empty(elems(pc_out, pc_in), dtype=torch.bfloat16, device=w.device)
        _lib.check(pack(_lib.dev_ptr(w, "w"), _lib.dev_ptr(packed, "packed", torch.bfloat16),
                        cout, cin, 1 if transpose_flip else 0, _lib.stream_ptr()),
                   "st_conv3x3_pack_weights" + _SUFFIX[compute])
        return packed
    assert compute == "f32", f"unknown compute dtype {compute!r}"
    packed = torch.empty(lib.st_conv3x3_packed_elems(pc_out, pc_in), dtype=torch.float32, device=w.device)
    _lib.check(lib.st_conv3x3_pack_weights(_lib.dev_ptr(w, "w"), _lib.dev_ptr(packed, "packed"), cout, cin,
                                           1 if transpose_flip else 0, _lib.stream_ptr()),
               "st_conv3x3_pack_weights")
    return packed


def conv3x3(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
            upsample2x: bool = False, pad_mode="replicate", relu: bool = True) -> torch.Tensor:
    """act(bias + conv3x3(pad1(up2x(x)))) on NCHW fp32 (generator.py:47-76, 89-117)."""
    lib = _lib.load()
    assert x.dim() == 4, "conv3x3 expects NCHW"
    n, cin, h, w = (int(v) for v in x.shape)
    bf16 = conv_family(w_packed, cout, cin)
    s = 2 if upsample2x else 1
    y = torch.empty((n, cout, h * s, w * s), dtype=torch.float32, device=x.device)
    ev = None
    if bf16 == "f32up2":
        assert upsample2x and _pad_code(pad_mode) == PAD_REPLICATE, "sub-pixel weights: an upsample layer, replicate pad"
        if KERNEL_TIMER.enabled:       # EXECUTED work: 4 classes x 4 taps per source pixel (36 in the direct form)
            ev = KERNEL_TIMER.record(conv_kernel_name(cout, bf16), 2.0 * n * h * w * cin * cout * 16,
                                     4.0 * (x.numel() + y.numel()))
        _lib.check(lib.st_conv3x3_up2_fwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(w_packed, "w_packed"),
                                          _lib.dev_ptr(bias, "bias") if bias is not None else None,
                                          _lib.dev_ptr(y, "y"), n, cin, h, w, cout, int(relu), _lib.stream_ptr()),
                   "st_conv3x3_up2_fwd")
        if ev is not None:
            ev.record()
        return y
    if KERNEL_TIMER.enabled:
        ev = KERNEL_TIMER.record(conv_kernel_name(cout, bf16), 2.0 * n * h * s * w * s * cin * cout * 9,
                                 4.0 * (x.numel() + y.numel()))
    fn = getattr(lib, "st_conv3x3_fwd" + _SUFFIX[bf16])
    _lib.check(fn(_lib.dev_ptr(x, "x"), _lib.dev_ptr(w_packed, "w_packed", w_packed.dtype),
                  _lib.dev_ptr(bias, "bias") if bias is not None else None,
                  _lib.dev_ptr(y, "y"), n, cin, h, w, cout, int(upsample2x),
                  _pad_code(pad_mode), int(relu), _lib.stream_ptr()), "st_conv3x3_fwd")
    if ev is not None:
        ev.record()
    return y


def conv3x3_pool(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
                 pad_mode="replicate", relu: bool = True) -> torch.Tensor:
    """maxpool2x2_ceil(act(bias + conv3x3(pad1(x)))) in one launch (generator.py:51-53, 58-60, 71-73); forward only."""
    lib = _lib.load()
    assert x.dim() == 4, "conv3x3_pool expects NCHW"
    n, cin, h, w = (int(v) for v in x.shape)
    bf16 = conv_family(w_packed, cout, cin)
    y = torch.empty((n, cout, (h + 1) // 2, (w + 1) // 2), dtype=torch.float32, device=x.device)
    ev = None
    if KERNEL_TIMER.enabled:
        ev = KERNEL_TIMER.record(conv_kernel_name(cout, bf16), 2.0 * n * h * w * cin * cout * 9,
                                 4.0 * (x.numel() + y.numel()))
    fn = getattr(lib, "st_conv3x3_pool_fwd" + _SUFFIX[bf16])
    _lib.check(fn(_lib.dev_ptr(x, "x"), _lib.dev_ptr(w_packed, "w_packed", w_packed.dtype),
                  _lib.dev_ptr(bias, "bias") if bias is not None else None, _lib.dev_ptr(y, "y"),
                  n, cin, h, w, cout, _pad_code(pad_mode), int(relu), _lib.stream_ptr()), "st_conv3x3_pool_fwd")
    if ev is not None:
        ev.record()
    return y


def to_g8(x: torch.Tensor) -> torch.Tensor:
    """fp32 NCHW -> the bf16 channel-group layout [n, c/8, h, w, 8] ("G8", include/st_hip.h) by torch ops
    (round-to-nearest-even, like the kernels); c % 8 == 0."""
    n, c, h, w = x.shape
    assert c % 8 == 0, f"G8 needs c % 8 == 0, got {c}"
    return x.view(n, c // 8, 8, h, w).permute(0, 1, 3, 4, 2).to(torch.bfloat16).contiguous()


def from_g8(g: torch.Tensor) -> torch.Tensor:
    """G8 bf16 [n, c/8, h, w, 8] -> fp32 NCHW (exact)."""
    n, cg, h, w, e = g.shape
    assert e == 8 and g.dtype == torch.bfloat16
    return g.permute(0, 1, 4, 2, 3).reshape(n, cg * 8, h, w).float().contiguous()


def is_g8(x: torch.Tensor) -> bool:
    return x.dim() == 5 and x.dtype == torch.bfloat16 and x.shape[-1] == 8


def to_g8_pair(x: torch.Tensor) -> torch.Tensor:
    """fp32 NCHW -> the bf16x3 hi/lo pair layout bf16 [n, c/8, 2, h, w, 8]: part 0 = bf16(v), part 1 = bf16(v - hi)."""
    n, c, h, w = x.shape
    assert c % 8 == 0, f"G8 needs c % 8 == 0, got {c}"
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    pair = torch.stack([hi, lo], dim=0).view(2, n, c // 8, 8, h, w)
    return pair.permute(1, 2, 0, 4, 5, 3).contiguous()


def from_g8_pair(g: torch.Tensor) -> torch.Tensor:
    """bf16x3 pair layout -> fp32 NCHW (hi + lo, exact in fp32)."""
    n, cg, two, h, w, e = g.shape
    assert two == 2 and e == 8 and g.dtype == torch.bfloat16
    v = g[:, :, 0].float() + g[:, :, 1].float()
    return v.permute(0, 1, 4, 2, 3).reshape(n, cg * 8, h, w).contiguous()


def is_g8_pair(x: torch.Tensor) -> bool:
    return x.dim() == 6 and x.dtype == torch.bfloat16 and x.shape[-1] == 8 and x.shape[2] == 2


def conv3x3_g8(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
               upsample2x: bool = False, pad_mode="replicate", relu: bool = True, pool: bool = False,
               y_g8: bool = True) -> torch.Tensor:
    """bf16-mode conv with the layer-to-layer activation layout G8 on either side (st_conv3x3_fwd_bf16_g8):
    ``x`` is fp32 NCHW or a G8 tensor (is_g8); the result is G8 when ``y_g8`` else fp32 NCHW.  Bit-identical
    to conv3x3 / conv3x3_pool with bf16-packed weights on the same values.  Forward only."""
    lib = _lib.load()
    x_g8 = is_g8(x) or is_g8_pair(x)
    if x_g8:
        n, cg, h, w = int(x.shape[0]), int(x.shape[1]), int(x.shape[-3]), int(x.shape[-2])
        cin = cg * 8
    else:
        assert x.dim() == 4 and x.dtype == torch.float32, "conv3x3_g8 expects fp32 NCHW or a G8 tensor"
        n, cin, h, w = (int(v) for v in x.shape)
    assert x.is_contiguous()
    fam = conv_family(w_packed, cout, cin)
    assert fam in ("bf16", "bf16x3"), "conv3x3_g8 needs bf16- or bf16x3-packed weights"
    pair = fam == "bf16x3"
    assert not x_g8 or is_g8_pair(x) == pair, "G8 input form does not match the packed weights (bf16 vs bf16x3)"
    s = 2 if upsample2x else 1
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if pool else (h * s, w * s)
    if y_g8:
        shape = (n, cout // 8, 2, ho, wo, 8) if pair else (n, cout // 8, ho, wo, 8)
        y = torch.empty(shape, dtype=torch.bfloat16, device=x.device)
    else:
        y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device)
    ev = None
    if KERNEL_TIMER.enabled:
        ev = KERNEL_TIMER.record(conv_kernel_name(cout, fam), 2.0 * n * h * s * w * s * cin * cout * 9,
                                 float(x.numel() * x.element_size() + y.numel() * y.element_size()))
    _lib.check(getattr(lib, "st_conv3x3_fwd" + _SUFFIX[fam] + "_g8")(
        _lib.dev_ptr(x, "x", x.dtype), int(x_g8), _lib.dev_ptr(w_packed, "w_packed", w_packed.dtype),
        _lib.dev_ptr(bias, "bias") if bias is not None else None, _lib.dev_ptr(y, "y", y.dtype), int(y_g8),
        n, cin, h, w, cout, int(upsample2x), _pad_code(pad_mode), int(relu), int(pool), _lib.stream_ptr()),
        "st_conv3x3_fwd_bf16_g8")
    if ev is not None:
        ev.record()
    return y


def conv3x3_pool_g8_codes(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
                          pad_mode="zero", relu: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """conv + ReLU + 2x2 max pool of a bf16-mode chain under autograd (st_conv3x3_pool_fwd_bf16_g8_codes): the pooled
    activation in G8 and the uint8 winner codes [n, cout/8, h/2, w/2, 8] that grad_pool_codes_g8 routes the gradient
    by.  ``x`` fp32 NCHW or G8; even h and w."""
    lib = _lib.load()
    x_g8 = is_g8(x)
    if x_g8:
        n, cin, h, w = int(x.shape[0]), int(x.shape[1]) * 8, int(x.shape[2]), int(x.shape[3])
    else:
        assert x.dim() == 4 and x.dtype == torch.float32
        n, cin, h, w = (int(v) for v in x.shape)
    assert x.is_contiguous() and h % 2 == 0 and w % 2 == 0 and cout % 8 == 0
    assert conv_family(w_packed, cout, cin) == "bf16", "conv3x3_pool_g8_codes needs bf16-packed weights"
    y = torch.empty((n, cout // 8, h // 2, w // 2, 8), dtype=torch.bfloat16, device=x.device)
    codes = torch.empty((n, cout // 8, h // 2, w // 2, 8), dtype=torch.uint8, device=x.device)
    ev = None
    if KERNEL_TIMER.enabled:
        ev = KERNEL_TIMER.record(conv_kernel_name(cout, "bf16"), 2.0 * n * h * w * cin * cout * 9,
                                 float(x.numel() * x.element_size() + y.numel() * 3))
    _lib.check(lib.st_conv3x3_pool_fwd_bf16_g8_codes(
        _lib.dev_ptr(x, "x", x.dtype), int(x_g8), _lib.dev_ptr(w_packed, "w_packed", w_packed.dtype),
        _lib.dev_ptr(bias, "bias") if bias is not None else None, _lib.dev_ptr(y, "y", torch.bfloat16),
        _lib.dev_ptr(codes, "codes", torch.uint8), n, cin, h, w, cout, _pad_code(pad_mode), int(relu),
        _lib.stream_ptr()), "st_conv3x3_pool_fwd_bf16_g8_codes")
    if ev is not None:
        ev.record()
    return y, codes


def conv3x3_g8_tap_pool(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
                        pad_mode="zero", relu: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """(fp32 NCHW activation, its 2x2 max pool in G8) of a bf16-mode conv with a G8 input, one launch
    (st_conv3x3_fwd_bf16_g8_tap_pool): a loss tap whose next layer is a pool.  Even h and w."""
    lib = _lib.load()
    assert is_g8(x) and x.is_contiguous()
    n, cin, h, w = int(x.shape[0]), int(x.shape[1]) * 8, int(x.shape[2]), int(x.shape[3])
    assert h % 2 == 0 and w % 2 == 0 and cout % 8 == 0
    assert conv_family(w_packed, cout, cin) == "bf16", "conv3x3_g8_tap_pool needs bf16-packed weights"
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    yp = torch.empty((n, cout // 8, h // 2, w // 2, 8), dtype=torch.bfloat16, device=x.device)
    ev = None
    if KERNEL_TIMER.enabled:
        ev = KERNEL_TIMER.record(conv_kernel_name(cout, "bf16"), 2.0 * n * h * w * cin * cout * 9,
                                 float(x.numel() * 2 + y.numel() * 4 + yp.numel() * 2))
    _lib.check(lib.st_conv3x3_fwd_bf16_g8_tap_pool(
        _lib.dev_ptr(x, "x", torch.bfloat16), _lib.dev_ptr(w_packed, "w_packed", w_packed.dtype),
        _lib.dev_ptr(bias, "bias") if bias is not None else None, _lib.dev_ptr(y, "y"),
        _lib.dev_ptr(yp, "y_pooled", torch.bfloat16), n, cin, h, w, cout, _pad_code(pad_mode), int(relu),
        _lib.stream_ptr()), "st_conv3x3_fwd_bf16_g8_tap_pool")
    if ev is not None:
        ev.record()
    return y, yp


def grad_pool_codes_g8(g_pooled: torch.Tensor, codes: torch.Tensor) -> torch.Tensor:
    """The G8 gradient w.r.t. a pooled activation routed through conv3x3_pool_g8_codes' winner codes
    (st_grad_pool_codes_g8): G8 [n, c/8, 2h, 2w, 8], ReLU gate included."""
    lib = _lib.load()
    assert is_g8(g_pooled) and codes.dtype == torch.uint8 and tuple(codes.shape) == tuple(g_pooled.shape)
    n, cg, ho, wo = (int(v) for v in g_pooled.shape[:4])
    out = torch.empty((n, cg, 2 * ho, 2 * wo, 8), dtype=torch.bfloat16, device=g_pooled.device)
    _lib.check(lib.st_grad_pool_codes_g8(_lib.dev_ptr(g_pooled, "g_pooled", torch.bfloat16),
                                         _lib.dev_ptr(codes, "codes", torch.uint8),
                                         _lib.dev_ptr(out, "out", torch.bfloat16), n, cg * 8, 2 * ho, 2 * wo,
                                         _lib.stream_ptr()), "st_grad_pool_codes_g8")
    return out


def conv3x3_dgrad(dy: torch.Tensor, y_gate: Optional[torch.Tensor], w_packed_t: torch.Tensor, cin: int,
                  upsample2x: bool = False, pad_mode="replicate",
                  dx_gate: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Gradient w.r.t. the input of the fused conv layer; ``y_gate`` = its post-ReLU output (or None);
    ``dx_gate`` (replicate padding only) = the layer's INPUT when that is a ReLU output: the result is
    zeroed where it is <= 0, i.e. it comes out already gated for the producing layer's backward."""
    lib = _lib.load()
    n, cout, h, w = (int(v) for v in dy.shape)
    pm = _pad_code(pad_mode)
    s = 2 if upsample2x else 1
    dx = torch.empty((n, cin, h // s, w // s), dtype=torch.float32, device=dy.device)
    assert dx_gate is None or tuple(dx_gate.shape) == tuple(dx.shape), "dx_gate must have the shape of dx"
    nbytes = lib.st_conv3x3_dgrad_workspace_bytes(n, cin, h, w, pm)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dy.device) if nbytes else None
    ev = None
    bf16 = conv_family(w_packed_t, cin, cout)
    if KERNEL_TIMER.enabled:
        ev = KERNEL_TIMER.record(conv_kernel_name(cin, bf16) + " dgrad", 2.0 * n * h * w * cin * cout * 9,
                                 4.0 * (dy.numel() + dx.numel()))
    fn = getattr(lib, "st_conv3x3_dgrad" + _SUFFIX[bf16])
    _lib.check(fn(_lib.dev_ptr(dy, "dy"),
                  _lib.dev_ptr(y_gate, "y_gate") if y_gate is not None else None,
                  _lib.dev_ptr(w_packed_t, "w_packed_t", w_packed_t.dtype), _lib.dev_ptr(dx, "dx"),
                  _lib.dev_ptr(ws, "workspace") if ws is not None else None,
                  _lib.dev_ptr(dx_gate, "dx_gate") if dx_gate is not None else None,
                  n, cout, h, w, cin, int(upsample2x), pm, _lib.stream_ptr()), "st_conv3x3_dgrad")
    if ev is not None:
        ev.record()
    return dx


def conv3x3_dgrad_g8(dy: torch.Tensor, w_packed_t: torch.Tensor, cin: int, dx_g8: bool,
                     dx_gate: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Data gradient of a zero-padded bf16-mode layer inside a G8 chain (st_conv3x3_dgrad_bf16_g8): ``dy`` is fp32
    NCHW or G8 (then ALREADY gated by the layer's ReLU); the result is G8 when ``dx_g8`` (optionally zeroed where the
    G8 tensor ``dx_gate`` -- the layer's input activation -- is <= 0) else fp32 NCHW."""
    lib = _lib.load()
    dy_g8 = is_g8(dy)
    if dy_g8:
        n, cout, h, w = int(dy.shape[0]), int(dy.shape[1]) * 8, int(dy.shape[2]), int(dy.shape[3])
    else:
        assert dy.dim() == 4 and dy.dtype == torch.float32
        n, cout, h, w = (int(v) for v in dy.shape)
    assert conv_family(w_packed_t, cin, cout) == "bf16", "conv3x3_dgrad_g8 needs bf16-packed (transposed) weights"
    if dx_g8:
        dx = torch.empty((n, cin // 8, h, w, 8), dtype=torch.bfloat16, device=dy.device)
    else:
        dx = torch.empty((n, cin, h, w), dtype=torch.float32, device=dy.device)
    assert dx_gate is None or (dx_g8 and is_g8(dx_gate) and tuple(dx_gate.shape) == tuple(dx.shape)), \
        "dx_gate must be a G8 tensor of dx's shape"
    ev = None
    if KERNEL_TIMER.enabled:
        ev = KERNEL_TIMER.record(conv_kernel_name(cin, "bf16") + " dgrad", 2.0 * n * h * w * cin * cout * 9,
                                 float(dy.numel() * dy.element_size() + dx.numel() * dx.element_size()))
    _lib.check(lib.st_conv3x3_dgrad_bf16_g8(
        _lib.dev_ptr(dy, "dy", dy.dtype), int(dy_g8), _lib.dev_ptr(w_packed_t, "w_packed_t", w_packed_t.dtype),
        _lib.dev_ptr(dx, "dx", dx.dtype), int(dx_g8),
        _lib.dev_ptr(dx_gate, "dx_gate", torch.bfloat16) if dx_gate is not None else None,
        n, cout, h, w, cin, _lib.stream_ptr()), "st_conv3x3_dgrad_bf16_g8")
    if ev is not None:
        ev.record()
    return dx


def conv3x3_dgrad_g8_chain(dy: torch.Tensor, w_packed_t: torch.Tensor, cin: int, upsample2x: bool,
                           dx_gate: Optional[torch.Tensor], want_f32: bool, want_g8: bool, want_g16: bool,
                           want_bias: bool):
    """Data gradient of a replicate-padded bf16-mode layer inside the decoder's G8 chain
    (st_conv3x3_dgrad_bf16_g8_chain): ``dy`` fp32 NCHW or G8 (already gated); ``dx_gate`` = the layer's input
    activation (fp32 NCHW or G8) or None.  Returns (dx_f32, dx_g8, dx_g16, bias_slabs), None where not requested."""
    lib = _lib.load()
    dy_g8 = is_g8(dy)
    if dy_g8:
        n, cout, h, w = int(dy.shape[0]), int(dy.shape[1]) * 8, int(dy.shape[2]), int(dy.shape[3])
    else:
        n, cout, h, w = (int(v) for v in dy.shape)
    assert conv_family(w_packed_t, cin, cout) == "bf16"
    s = 2 if upsample2x else 1
    hs, ws = h // s, w // s
    dev = dy.device
    wsb = lib.st_conv3x3_dgrad_workspace_bytes(n, cin, h, w, PAD_REPLICATE)
    work = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
    dx = torch.empty((n, cin, hs, ws), dtype=torch.float32, device=dev) if want_f32 else None
    g8 = torch.empty((n, cin // 8, hs, ws, 8), dtype=torch.bfloat16, device=dev) if want_g8 else None
    g16 = torch.empty((n, cin, hs, ws), dtype=torch.bfloat16, device=dev) if want_g16 else None
    bias = None
    if want_bias:
        slabs = lib.st_conv3x3_dgrad_chain_bias_slabs(n, cin, h, w, int(upsample2x))
        assert slabs > 0
        bias = torch.empty((slabs, cin), dtype=torch.float32, device=dev)
    gate_g8 = dx_gate is not None and is_g8(dx_gate)
    if dx_gate is not None:
        assert tuple(dx_gate.shape) == ((n, cin // 8, hs, ws, 8) if gate_g8 else (n, cin, hs, ws)), "dx_gate shape"
    ev = None
    if KERNEL_TIMER.enabled:
        ev = KERNEL_TIMER.record(conv_kernel_name(cin, "bf16") + " dgrad", 2.0 * n * h * w * cin * cout * 9,
                                 float(dy.numel() * dy.element_size() + 4.0 * n * cin * hs * ws))
    ptr = lambda t, dt=torch.float32: _lib.dev_ptr(t, "tensor", dt) if t is not None else None      # noqa: E731
    _lib.check(lib.st_conv3x3_dgrad_bf16_g8_chain(
        _lib.dev_ptr(dy, "dy", dy.dtype), int(dy_g8), _lib.dev_ptr(w_packed_t, "w_packed_t", w_packed_t.dtype),
        _lib.dev_ptr(work, "workspace"), ptr(dx_gate, dx_gate.dtype if dx_gate is not None else None), int(gate_g8),
        ptr(dx), ptr(g8, torch.bfloat16), ptr(g16, torch.bfloat16), ptr(bias), n, cout, h, w, cin, int(upsample2x),
        _lib.stream_ptr()), "st_conv3x3_dgrad_bf16_g8_chain")
    if ev is not None:
        ev.record()
    return dx, g8, g16, bias


def conv3x3_wgrad_pre(x: torch.Tensor, g16: torch.Tensor, bias_slabs: Optional[torch.Tensor], upsample2x: bool,
                      pad_mode="replicate", dw_out: Optional[torch.Tensor] = None,
                      db_out: Optional[torch.Tensor] = None):
    """(dW, dbias) of a bf16-mode layer from the READY operand ``g16`` (bf16 [n, cout, H, W], gated) and the bias
    partial sums ``bias_slabs`` [slabs, cout] (st_conv3x3_wgrad_bf16_pre); ``x`` fp32 NCHW or G8."""
    lib = _lib.load()
    x_g8 = is_g8(x)
    if x_g8:
        n, cin, h, w = int(x.shape[0]), int(x.shape[1]) * 8, int(x.shape[2]), int(x.shape[3])
    else:
        n, cin, h, w = (int(v) for v in x.shape)
    cout = int(g16.shape[1])
    s = 2 if upsample2x else 1
    assert tuple(g16.shape) == (n, cout, h * s, w * s) and g16.dtype == torch.bfloat16
    dw = dw_out if dw_out is not None else torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    assert tuple(dw.shape) == (cout, cin, 3, 3)
    db = None
    if bias_slabs is not None:
        db = db_out if db_out is not None else torch.empty((cout,), dtype=torch.float32, device=x.device)
        assert tuple(db.shape) == (cout,) and bias_slabs.shape[1] == cout
    nbytes = lib.st_conv3x3_wgrad_bf16_pre_workspace_bytes(n, cout, cin, h * s, w * s)
    assert nbytes > 0, "conv3x3_wgrad_pre: shape outside the staged weight-gradient form"
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    ev = None
    if KERNEL_TIMER.enabled:
        ev = KERNEL_TIMER.record("conv3x3_wgrad_bf16_kernel", 2.0 * n * h * s * w * s * cin * cout * 9,
                                 2.0 * g16.numel() + x.numel() * x.element_size())
    _lib.check(lib.st_conv3x3_wgrad_bf16_pre(
        _lib.dev_ptr(x, "x", x.dtype), int(x_g8), _lib.dev_ptr(g16, "g16", torch.bfloat16),
        _lib.dev_ptr(bias_slabs, "bias_slabs") if bias_slabs is not None else None,
        int(bias_slabs.shape[0]) if bias_slabs is not None else 0, _lib.dev_ptr(dw, "dw"),
        _lib.dev_ptr(db, "db") if db is not None else None, _lib.dev_ptr(ws, "workspace"), n, cin, h, w, cout,
        int(upsample2x), _pad_code(pad_mode), _lib.stream_ptr()), "st_conv3x3_wgrad_bf16_pre")
    if ev is not None:
        ev.record()
    return dw, db


def conv3x3_wgrad_cout3_g8(x_g8: torch.Tensor, dy: torch.Tensor, upsample2x: bool = False, pad_mode="replicate",
                           dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None):
    """(dW [3, cin, 3, 3], dbias [3]) of the 3-output-channel top layer from its G8 input and the fp32 ``dy``
    (st_conv3x3_wgrad_cout3_xg8: the vector-ALU weight gradient reading the bf16 activation directly)."""
    lib = _lib.load()
    assert is_g8(x_g8)
    n, cin, h, w = int(x_g8.shape[0]), int(x_g8.shape[1]) * 8, int(x_g8.shape[2]), int(x_g8.shape[3])
    s = 2 if upsample2x else 1
    assert tuple(dy.shape) == (n, 3, h * s, w * s) and dy.dtype == torch.float32, f"dy shape {tuple(dy.shape)}"
    dw = dw_out if dw_out is not None else torch.empty((3, cin, 3, 3), dtype=torch.float32, device=dy.device)
    db = db_out if db_out is not None else torch.empty((3,), dtype=torch.float32, device=dy.device)
    assert tuple(dw.shape) == (3, cin, 3, 3) and tuple(db.shape) == (3,)
    ws = torch.empty(lib.st_conv3x3_wgrad_workspace_bytes(n, 3, cin, h * s, w * s) // 4, dtype=torch.float32,
                     device=dy.device)
    _lib.check(lib.st_conv3x3_wgrad_cout3_xg8(
        _lib.dev_ptr(x_g8, "x_g8", torch.bfloat16), _lib.dev_ptr(dy, "dy"), _lib.dev_ptr(dw, "dw"), _lib.dev_ptr(db, "db"),
        _lib.dev_ptr(ws, "workspace"), n, cin, h, w, int(upsample2x), _pad_code(pad_mode), _lib.stream_ptr()),
        "st_conv3x3_wgrad_cout3_xg8")
    return dw, db


def wgrad_pre_ok(n: int, cout: int, cin: int, h: int, w: int) -> bool:
    """Is [n, cout, h, w] (conv-input dims after the upsample) a shape of the staged weight-gradient form?"""
    return _lib.load().st_conv3x3_wgrad_bf16_pre_workspace_bytes(n, cout, cin, h, w) > 0


def grad_join_g8(shape, chain: Optional[torch.Tensor] = None, chain_pooled: Optional[torch.Tensor] = None,
                 x_gate: Optional[torch.Tensor] = None, add_post: Optional[torch.Tensor] = None,
                 add_pre: Optional[torch.Tensor] = None, aff_x: Optional[torch.Tensor] = None,
                 aff_post: Optional[torch.Tensor] = None, aff_pre: Optional[torch.Tensor] = None) -> torch.Tensor:
    """G8 bf16 of ``[x_gate > 0] * (chain + add_post) + add_pre`` for an fp32 NCHW ``shape`` (st_grad_join_g8);
    ``chain_pooled`` instead of ``chain``: the gradient w.r.t. maxpool2x2_floor(x_gate), routed through the pool.
    ``aff_post`` / ``aff_pre``: a loss gradient as per-plane coefficients [n * c, 4] (meanstd_loss_coeffs) of the tap
    ``aff_x`` in place of the tensor ``add_post`` / ``add_pre`` (st_grad_join_g8_affine)."""
    lib = _lib.load()
    n, c, h, w = (int(v) for v in shape)
    ref = next(t for t in (chain, chain_pooled, x_gate, add_post, add_pre, aff_x) if t is not None)
    for name, t in (("chain", chain), ("x_gate", x_gate), ("add_post", add_post), ("add_pre", add_pre), ("aff_x", aff_x)):
        assert t is None or tuple(t.shape) == (n, c, h, w), f"{name}: shape {tuple(t.shape)} != {(n, c, h, w)}"
    assert chain_pooled is None or tuple(chain_pooled.shape) == (n, c, h // 2, w // 2)
    for name, t in (("aff_post", aff_post), ("aff_pre", aff_pre)):
        assert t is None or (tuple(t.shape) == (n * c, 4) and aff_x is not None), f"{name}: [n * c, 4] coefficients + aff_x"
    out = torch.empty((n, c // 8, h, w, 8), dtype=torch.bfloat16, device=ref.device)
    ptr = lambda t, name: _lib.dev_ptr(t, name) if t is not None else None      # noqa: E731
    if aff_post is None and aff_pre is None:
        _lib.check(lib.st_grad_join_g8(ptr(chain, "chain"), ptr(chain_pooled, "chain_pooled"), ptr(x_gate, "x_gate"),
                                       ptr(add_post, "add_post"), ptr(add_pre, "add_pre"),
                                       _lib.dev_ptr(out, "out", torch.bfloat16), n, c, h, w, _lib.stream_ptr()),
                   "st_grad_join_g8")
    else:
        _lib.check(lib.st_grad_join_g8_affine(ptr(chain, "chain"), ptr(chain_pooled, "chain_pooled"), ptr(x_gate, "x_gate"),
                                              ptr(add_post, "add_post"), ptr(add_pre, "add_pre"), ptr(aff_x, "aff_x"),
                                              ptr(aff_post, "aff_post"), ptr(aff_pre, "aff_pre"),
                                              _lib.dev_ptr(out, "out", torch.bfloat16), n, c, h, w, _lib.stream_ptr()),
                   "st_grad_join_g8_affine")
    return out


def meanstd_loss_coeffs(ma, sa, mb, sb, gout: torch.Tensor, hw: int) -> torch.Tensor:
    """[planes, 4] = (k0, k1, mean, 0) with d AdaINStyleLoss / d x = k0 + k1 * (x - mean) per (image, channel) plane
    (st_meanstd_loss_bwd_coeffs: st_meanstd_loss_bwd's own per-plane scalars)."""
    lib = _lib.load()
    planes = int(ma.numel())
    coeffs = torch.empty((planes, 4), dtype=torch.float32, device=ma.device)
    g = gout.contiguous().reshape(1)
    _lib.check(lib.st_meanstd_loss_bwd_coeffs(_lib.dev_ptr(ma, "ma"), _lib.dev_ptr(sa, "sa"), _lib.dev_ptr(mb, "mb"),
                                              _lib.dev_ptr(sb, "sb"), _lib.dev_ptr(g, "gout"), 1.0,
                                              _lib.dev_ptr(coeffs, "coeffs"), planes, int(hw), _lib.stream_ptr()),
               "st_meanstd_loss_bwd_coeffs")
    return coeffs


def conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, y_gate: Optional[torch.Tensor], want_bias: bool = True,
                  upsample2x: bool = False, pad_mode="replicate", compute: str = "f32",
                  dw_out: Optional[torch.Tensor] = None,
                  db_out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """(dW [cout,cin,3,3], dbias [cout] or None) of the fused conv layer; ``compute="bf16"`` rounds both
    GEMM operands (gated dy and the padded input) to bf16, accumulation and outputs stay fp32;
    "bf16x3" uses the exact fp32 kernel (precision.py).  ``dw_out`` / ``db_out``: contiguous fp32 tensors the
    kernel OVERWRITES instead of fresh allocations (the trainer passes slices of its flat gradient bucket)."""
    if compute == "bf16x3":
        compute = "f32"
    lib = _lib.load()
    n, cin, h, w = (int(v) for v in x.shape)
    cout = int(dy.shape[1])
    s = 2 if upsample2x else 1
    assert tuple(dy.shape) == (n, cout, h * s, w * s), f"dy shape {tuple(dy.shape)} does not match x {tuple(x.shape)}"
    if dw_out is not None:
        assert tuple(dw_out.shape) == (cout, cin, 3, 3), f"dw_out shape {tuple(dw_out.shape)}"
        dw = dw_out
    else:
        dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    if want_bias and db_out is not None:
        assert tuple(db_out.shape) == (cout,), f"db_out shape {tuple(db_out.shape)}"
        db = db_out
    else:
        db = torch.empty((cout,), dtype=torch.float32, device=x.device) if want_bias else None
    ws_bytes = lib.st_conv3x3_wgrad_bf16_workspace_bytes if compute == "bf16" else lib.st_conv3x3_wgrad_workspace_bytes
    ws = torch.empty(ws_bytes(n, cout, cin, h * s, w * s) // 4, dtype=torch.float32, device=x.device)
    ev = None
    if KERNEL_TIMER.enabled:
        ev = KERNEL_TIMER.record("conv3x3_wgrad_bf16_kernel" if compute == "bf16" else "conv3x3_wgrad_kernel",
                                 2.0 * n * h * s * w * s * cin * cout * 9, 4.0 * (dy.numel() + x.numel()))
    fn = lib.st_conv3x3_wgrad_bf16 if compute == "bf16" else lib.st_conv3x3_wgrad
    _lib.check(fn(_lib.dev_ptr(x, "x"), _lib.dev_ptr(dy, "dy"),
                                    _lib.dev_ptr(y_gate, "y_gate") if y_gate is not None else None,
                                    _lib.dev_ptr(dw, "dw"), _lib.dev_ptr(db, "db") if db is not None else None,
                                    _lib.dev_ptr(ws, "workspace"), n, cin, h, w, cout, int(upsample2x),
                                    _pad_code(pad_mode), _lib.stream_ptr()), "st_conv3x3_wgrad")
    if ev is not None:
        ev.record()
    return dw, db


def maxpool2x2_bwd(x: torch.Tensor, dy: torch.Tensor, ceil_mode: bool) -> torch.Tensor:
    lib = _lib.load()
    n, c, h, w = (int(v) for v in x.shape)
    dx = torch.empty_like(x)
    _lib.check(lib.st_maxpool2x2_bwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(dy, "dy"), _lib.dev_ptr(dx, "dx"), n * c, h, w,
                                     int(ceil_mode), _lib.stream_ptr()), "st_maxpool2x2_bwd")
    return dx


def sympad(x: torch.Tensor, padding=(1, 1, 1, 1)) -> torch.Tensor:
    """SymmetricPadding2D.forward (generator.py:19-39); padding = (left, right, top, bottom)."""
    lib = _lib.load()
    left, right, top, bottom = (int(v) for v in padding)
    h, w = int(x.shape[-2]), int(x.shape[-1])
    nc = x.numel() // (h * w)
    y = torch.empty(tuple(x.shape[:-2]) + (h + top + bottom, w + left + right), dtype=torch.float32,
                    device=x.device)
    _lib.check(lib.st_sympad_fwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(y, "y"), nc, h, w, left, right, top, bottom,
                                 _lib.stream_ptr()), "st_sympad_fwd")
    return y


def conv1x1_3to3(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.Conv2d(3, 3, (1, 1)) on the raw image (generator.py:46)."""
    lib = _lib.load()
    assert x.dim() == 4 and x.shape[1] == 3 and w.numel() == 9
    n, _, h, wd = (int(v) for v in x.shape)
    y = torch.empty_like(x)
    _lib.check(lib.st_conv1x1_3to3_fwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(w.detach(), "w"),
                                       _lib.dev_ptr(b.detach(), "b") if b is not None else None,
                                       _lib.dev_ptr(y, "y"), n, h * wd, _lib.stream_ptr()), "st_conv1x1_3to3_fwd")
    return y


def normalize(x: torch.Tensor, mean, std) -> torch.Tensor:
    """transforms.Normalize(mean, std) (trainer.py:299); mean=None gives x / std (its backward)."""
    import ctypes
    lib = _lib.load()
    n, c, h, w = (int(v) for v in x.shape)
    y = torch.empty_like(x)
    arr = ctypes.c_float * c
    m = arr(*[float(v) for v in mean]) if mean is not None else None
    s = arr(*[float(v) for v in std])
    _lib.check(lib.st_normalize_fwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(y, "y"), n, c, h * w,
                                    ctypes.cast(m, ctypes.c_void_p) if m is not None else None,
                                    ctypes.cast(s, ctypes.c_void_p), _lib.stream_ptr()), "st_normalize_fwd")
    return y


def maxpool2x2(x: torch.Tensor, ceil_mode: bool) -> torch.Tensor:
    """MaxPool2d((2,2),(2,2),(0,0),ceil_mode) (generator.py:53,60,73) / floor (torchvision vgg19)."""
    lib = _lib.load()
    n, c, h, w = (int(v) for v in x.shape)
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if ceil_mode else (h // 2, w // 2)
    y = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
    _lib.check(lib.st_maxpool2x2_fwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(y, "y"), n * c, h, w, int(ceil_mode),
                                     _lib.stream_ptr()), "st_maxpool2x2_fwd")
    return y


def maxpool2x2_g8(x: torch.Tensor) -> torch.Tensor:
    """Floor-mode 2x2 max pool of fp32 NCHW written in G8 (st_maxpool2x2_fwd_g8); c % 8 == 0, even h and w."""
    lib = _lib.load()
    n, c, h, w = (int(v) for v in x.shape)
    y = torch.empty((n, c // 8, h // 2, w // 2, 8), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.st_maxpool2x2_fwd_g8(_lib.dev_ptr(x, "x"), _lib.dev_ptr(y, "y", torch.bfloat16), n, c, h, w,
                                        _lib.stream_ptr()), "st_maxpool2x2_fwd_g8")
    return y


def adain(content: torch.Tensor, style: torch.Tensor, eps_content: float, eps_style: float,
          alpha: float = 1.0, save_stats: bool = False,
          eps_dev: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """alpha * AdaIN(content, style) + (1 - alpha) * content (transformer.py:46-54, trainer.py:702).
    ``eps_dev``: optional 2-element device tensor (eps_content, eps_style) read by the kernel instead of the
    two host floats -- no host read of the trainable eps, no synchronisation."""
    lib = _lib.load()
    assert content.dim() == 4 and style.dim() == 4
    assert content.shape[:2] == style.shape[:2]          # transformer.py:47
    n, c, hc, wc = (int(v) for v in content.shape)
    hs, ws = int(style.shape[2]), int(style.shape[3])
    out = torch.empty_like(content)
    stats = torch.empty((n * c, 4), dtype=torch.float32, device=content.device) if save_stats else None
    ev = None
    if KERNEL_TIMER.enabled:   # algorithmic bytes: read content once, read style once, write out once
        ev = KERNEL_TIMER.record("adain_fwd", 0.0, 4.0 * (2 * content.numel() + style.numel()))
    if eps_dev is not None:
        assert eps_dev.numel() == 2, "eps_dev holds (eps_content, eps_style)"
        _lib.check(lib.st_adain_fwd_dev_eps(_lib.dev_ptr(content, "content"), _lib.dev_ptr(style, "style"),
                                            _lib.dev_ptr(out, "out"), n, c, hc * wc, hs * ws,
                                            _lib.dev_ptr(eps_dev, "eps_dev"), float(alpha),
                                            _lib.dev_ptr(stats, "stats") if stats is not None else None,
                                            _lib.stream_ptr()), "st_adain_fwd_dev_eps")
    else:
        _lib.check(lib.st_adain_fwd(_lib.dev_ptr(content, "content"), _lib.dev_ptr(style, "style"),
                                    _lib.dev_ptr(out, "out"), n, c, hc * wc, hs * ws, float(eps_content),
                                    float(eps_style), float(alpha),
                                    _lib.dev_ptr(stats, "stats") if stats is not None else None,
                                    _lib.stream_ptr()), "st_adain_fwd")
    if ev is not None:
        ev.record()
    return out, stats


def mean_std(x: torch.Tensor, eps: float = 1e-5) -> Tuple[torch.Tensor, torch.Tensor]:
    """calc_mean_std (transformer.py:15-24 / losses.py:20-28): mean and sqrt(unbiased var + eps), [N,C,1,1]."""
    lib = _lib.load()
    assert x.dim() == 4
    n, c, h, w = (int(v) for v in x.shape)
    mean = torch.empty((n, c, 1, 1), dtype=torch.float32, device=x.device)
    std = torch.empty((n, c, 1, 1), dtype=torch.float32, device=x.device)
    _lib.check(lib.st_meanstd_fwd(_lib.dev_ptr(x, "x"), _lib.dev_ptr(mean, "mean"), _lib.dev_ptr(std, "std"),
                                  n * c, h * w, float(eps), _lib.stream_ptr()), "st_meanstd_fwd")
    return mean, std


# ---- batched C x C products (csrc/bmm.hip) --------------------------------------------------------------------
def bmm_ok(a: torch.Tensor) -> bool:
    """Shapes st_bmm_f32 takes: [batch, n, n] fp32 on the GPU with n % 64 == 0 (every VGG19 channel count)."""
    return a.dim() == 3 and a.shape[-1] == a.shape[-2] and int(a.shape[-1]) % 64 == 0 and a.dtype == torch.float32 and a.is_cuda


def bmm_raw(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False, alpha: float = 1.0,
            diag: float = 0.0) -> torch.Tensor:
    """``alpha * op(a) @ op(b) + diag * I`` for [batch, n, n] fp32 tensors (st_bmm_f32, fp32 MFMA); no autograd."""
    lib = _lib.load()
    a, b = a.contiguous(), b.contiguous()
    assert bmm_ok(a) and tuple(a.shape) == tuple(b.shape), f"bmm: {tuple(a.shape)} x {tuple(b.shape)}"
    c = torch.empty_like(a)
    _lib.check(lib.st_bmm_f32(_lib.dev_ptr(a, "a"), _lib.dev_ptr(b, "b"), _lib.dev_ptr(c, "c"), int(a.shape[0]),
                              int(a.shape[1]), int(trans_a), int(trans_b), float(alpha), float(diag), _lib.stream_ptr()),
               "st_bmm_f32")
    return c


class _BmmFunction(torch.autograd.Function):
    """c = op(a) @ op(b) with dA, dB by the same kernel (the four transpose combinations of a GEMM's adjoints)."""

    @staticmethod
    def forward(ctx, a, b, trans_a, trans_b):
        ctx.save_for_backward(a, b)
        ctx.ta, ctx.tb = trans_a, trans_b
        return bmm_raw(a, b, trans_a, trans_b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ta, tb = ctx.ta, ctx.tb
        g = g.contiguous()
        ga = gb = None
        if ctx.needs_input_grad[0]:       # d op(a) = g @ op(b)^T; a stored transposed: d a = (g @ op(b)^T)^T = op(b) @ g^T
            ga = bmm_raw(b, g, tb, True) if ta else bmm_raw(g, b, False, not tb)
        if ctx.needs_input_grad[1]:       # d op(b) = op(a)^T @ g; b stored transposed: d b = g^T @ op(a)
            gb = bmm_raw(g, a, True, ta) if tb else bmm_raw(a, g, not ta, False)
        return ga, gb, None, None


def bmm(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False) -> torch.Tensor:
    """``op(a) @ op(b)`` for [batch, n, n] fp32 tensors, differentiable (st_bmm_f32 forward and backward)."""
    return _BmmFunction.apply(a.contiguous(), b.contiguous(), bool(trans_a), bool(trans_b))
