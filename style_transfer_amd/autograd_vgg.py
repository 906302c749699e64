"""The loss VGG's pass over the DECODED image in the bf16 compute mode as one autograd node.

trainer.py:446-457 runs the content and the style extractor (torchvision vgg19.features under an
IntermediateLayerGetter, trainer.py:165-199) on the decoded image and loss.backward() (trainer.py:472) pushes
the loss gradients back through them to the decoder.  The network is frozen: only data gradients are needed.

Forward: the same kernels and hand-overs as the no-gradient pass (loss_network._forward_multi_g8): a conv whose
output nobody captures hands it to the next conv in the bf16 channel-group layout G8; captured tensors (the loss
taps) and pool inputs stay fp32 NCHW.  Bit-identical to the per-layer fp32-activation form.

Backward: one walk down the layers.  The gradient travels in G8 as well: every data-gradient launch
(ops.conv3x3_dgrad_g8) takes a G8 operand that already carries its layer's ReLU gate and, where the layer below is
uncaptured, writes its result in G8 gated by that layer's (G8) activation; at a captured layer or a pool input ONE
kernel (ops.grad_join_g8) forms [y > 0] * (chain gradient routed through the pool + loss gradient) + pre-ReLU loss
gradient and writes the G8 operand.  Same values as the per-layer autograd graph (max-pool backward, fp32
accumulation, ReLU backward, then the bf16 rounding inside the data-gradient kernel's staging) without its fp32
round trips: no separate pool-backward, accumulate or ReLU-backward kernels, half the bytes per hand-over.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple

import torch

from . import ops


# ---- loss gradients handed over as per-plane affine coefficients -------------------------------------------------------
# A loss whose gradient w.r.t. a tap is affine in the tap per (image, channel) plane (AdaINStyleLoss) returns, from its
# autograd backward, a PLACEHOLDER of the tap's shape -- one NaN element expanded with stride 0, nothing materialised --
# and leaves the coefficients here under the placeholder's address; VGGChainFunction.backward recognises the
# placeholder and passes the coefficients to the join kernel.  Only taps with exactly ONE consumer are marked for this
# (``_st_chain_tap``, set in loss_network._forward_multi_chain); should autograd ever add the placeholder to another
# gradient, the NaN makes that visible at once instead of silently dropping a term.
_AFFINE: Dict[int, torch.Tensor] = {}
AFFINE_TAPS = True          # ST_NO_AFFINE_TAPS=1 (read in loss_network) keeps the tensor hand-over
# pool-input layers nobody captures: conv + ReLU + pool in one launch, winner codes for the backward
# (ops.conv3x3_pool_g8_codes); ST_NO_POOL_CODES=1 keeps the fp32 activation + maxpool2x2_g8 + pooled join (A/B, tests)
POOL_CODES = os.environ.get("ST_NO_POOL_CODES") is None
# captured pool-input layers (loss taps): the fp32 tap and the pooled G8 input of the next conv from one launch
# (ops.conv3x3_g8_tap_pool); ST_NO_TAP_POOL=1 keeps conv -> st_maxpool2x2_fwd_g8
TAP_POOL = os.environ.get("ST_NO_TAP_POOL") is None


def lazy_affine_grad(tap: torch.Tensor, coeffs: torch.Tensor) -> torch.Tensor:
    holder = torch.full((1,), float("nan"), dtype=tap.dtype, device=tap.device)
    if len(_AFFINE) > 256:      # placeholders no chain backward collected (a tap used outside the chain): drop them
        _AFFINE.clear()
    # the registry keeps ``holder`` alive, so no other tensor can sit at its address while the entry exists
    _AFFINE[holder.data_ptr()] = (holder, coeffs)
    return holder.expand(tap.shape)


def take_affine(g):
    """(coefficients, None) if ``g`` is a lazy_affine_grad placeholder, else (None, contiguous g)."""
    if g is None:
        return None, None
    if g.dim() == 4 and g.stride() == (0, 0, 0, 0) and g.data_ptr() in _AFFINE:
        return _AFFINE.pop(g.data_ptr())[1], None
    return None, g.contiguous()


def plan_units(layout, last_all: int) -> List[Tuple]:
    """[("conv", idx, has_relu) | ("pool", idx)] for layers 0..last_all (a conv's in-place ReLU is part of its unit)."""
    units, idx = [], 0
    while idx <= last_all:
        if layout[idx][0] == "conv":
            cont = idx + 1 <= last_all
            units.append(("conv", idx, cont))
            idx += 2 if cont else 1
        else:
            units.append(("pool", idx))
            idx += 1
    return units


def capture_sets(layout, sets: Sequence[Sequence[int]]):
    """(P, Q): conv indices captured BEFORE their ReLU (the last index of a set, when it is a conv) and indices
    captured after it (everything else; a conv index not last in its set has been overwritten by its in-place ReLU)."""
    pre = {max(s) for s in sets if layout[max(s)][0] == "conv"}
    post = {i for s in sets for i in s if not (i == max(s) and layout[i][0] == "conv")}
    return pre, post


class VGGChainFunction(torch.autograd.Function):
    """forward(x, net, last_all, P, Q) -> one fp32 tensor per key of ``keys(layout, last_all, P, Q)``, in that order
    ((P, Q) = capture_sets(layout, sets)); backward: see the module docstring."""

    @staticmethod
    def keys(layout, last_all, pre, post) -> List[Tuple[str, int]]:
        out = []
        for u in plan_units(layout, last_all):
            if u[0] == "conv":
                idx, cont = u[1], u[2]
                if idx in pre:
                    out.append(("pre", idx))
                if cont and (idx in post or idx + 1 in post):
                    out.append(("post", idx))
            elif u[1] in post:
                out.append(("post", u[1]))
        return out

    @staticmethod
    def forward(ctx, x, net, last_all, pre, post):
        units = plan_units(net.layout, last_all)
        outs: Dict[Tuple[str, int], torch.Tensor] = {}
        saved: List[torch.Tensor] = []
        meta = []                       # per unit: dict(kind, idx, ..., slot of the saved tensor)
        cur = x.contiguous()
        fused_pool = False              # the conv unit before this pool unit already pooled (winner codes saved)
        pooled_ready = None             # ... or wrote the pooled G8 tensor next to its fp32 tap
        for k, u in enumerate(units):
            if u[0] == "pool":
                idx = u[1]
                if fused_pool:
                    fused_pool = False
                    meta.append({"kind": "pool", "idx": idx, "fused": True})
                    continue
                c, h, w = int(cur.shape[1]), int(cur.shape[2]), int(cur.shape[3])
                if pooled_ready is not None:
                    cur, pooled_ready = pooled_ready, None
                    meta.append({"kind": "pool", "idx": idx, "h": h, "w": w})
                    continue
                if idx not in post and k + 1 < len(units) and h % 2 == 0 and w % 2 == 0 and c % 16 == 0:
                    cur = ops.maxpool2x2_g8(cur)         # nobody captures it: straight into the next conv's input form
                else:
                    cur = ops.maxpool2x2(cur, False)
                if idx in post:
                    outs[("post", idx)] = cur
                meta.append({"kind": "pool", "idx": idx, "h": h, "w": w})
                continue
            _, idx, cont = u
            conv = net.convs[str(idx)]
            cout, cin = int(conv.weight.shape[0]), int(conv.weight.shape[1])
            wp = net._packed.get(idx, conv.weight, False, "bf16")
            bias = conv.bias.detach()
            has_next = k + 1 < len(units)
            pool_next = has_next and units[k + 1][0] == "pool"
            m = {"kind": "conv", "idx": idx, "relu": cont, "cin": cin, "cout": cout, "g8": False}
            if idx in pre:
                y = ops.conv3x3_g8(cur, wp, bias, cout, False, ops.PAD_ZERO, relu=False, pool=False, y_g8=False)
                outs[("pre", idx)] = y
                cur = y
                if cont:
                    cur = torch.relu(y)
                    if idx in post or idx + 1 in post:
                        outs[("post", idx)] = cur
            else:
                assert cont, "a conv that ends the network is captured before its ReLU"
                captured = idx in post or idx + 1 in post
                y_g8 = not captured and not pool_next and has_next and cout % 8 == 0
                hh, ww = int(cur.shape[2]), int(cur.shape[3])
                if (POOL_CODES and pool_next and not captured and units[k + 1][1] not in post and k + 2 < len(units)
                        and cout % 16 == 0 and hh % 2 == 0 and ww % 2 == 0 and (ops.is_g8(cur) or cin % 16 == 0)):
                    cur, codes = ops.conv3x3_pool_g8_codes(cur, wp, bias, cout, ops.PAD_ZERO, relu=True)
                    m["codes"], m["shape"], m["slot"] = True, (int(x.shape[0]), cout, hh, ww), len(saved)
                    saved.append(codes)
                    meta.append(m)
                    fused_pool = True
                    continue
                if (TAP_POOL and captured and pool_next and units[k + 1][1] not in post and k + 2 < len(units)
                        and cout % 16 == 0 and hh % 2 == 0 and ww % 2 == 0 and ops.is_g8(cur)):
                    y, pooled_ready = ops.conv3x3_g8_tap_pool(cur, wp, bias, cout, ops.PAD_ZERO, relu=True)
                else:
                    y = ops.conv3x3_g8(cur, wp, bias, cout, False, ops.PAD_ZERO, relu=True, pool=False, y_g8=y_g8)
                m["g8"] = y_g8
                if captured:
                    outs[("post", idx)] = y
                cur = y
            m["shape"] = (int(x.shape[0]), cout, int(y.shape[-3] if m["g8"] else y.shape[-2]),
                          int(y.shape[-2] if m["g8"] else y.shape[-1]))
            m["slot"] = len(saved)
            saved.append(y)             # the gate of this unit (G8 or fp32; pre- or post-ReLU: same sign pattern)
            meta.append(m)
        keys = VGGChainFunction.keys(net.layout, last_all, pre, post)
        ctx.net, ctx.meta, ctx.out_keys, ctx.x_shape = net, meta, keys, tuple(x.shape)
        ctx.pre_idx = frozenset(pre)
        ctx.save_for_backward(*saved)
        return tuple(outs[k] for k in keys)

    @staticmethod
    def backward(ctx, *gouts):
        net, meta = ctx.net, ctx.meta
        saved = ctx.saved_tensors
        grads, affs = {}, {}
        for key, gout in zip(ctx.out_keys, gouts):
            affs[key], grads[key] = take_affine(gout)
        kind, g = None, None            # gradient w.r.t. the OUTPUT of the unit being visited: "f32" | "g8" | "pooled"
        dx = None
        for k in range(len(meta) - 1, -1, -1):
            m = meta[k]
            if m["kind"] == "pool":
                if m.get("fused"):
                    assert kind in (None, "g8")         # the launch above wrote the pooled gradient in G8, ungated
                    kind = "pooled_g8" if g is not None else None
                    continue
                assert kind in (None, "f32")
                tap = grads.get(("post", m["idx"]))
                assert affs.get(("post", m["idx"])) is None, "pool outputs are never marked for the affine hand-over"
                if g is not None and tap is not None:
                    g = g + tap
                elif tap is not None:
                    g = tap
                kind = "pooled" if g is not None else None
                continue
            idx = m["idx"]
            y = saved[m["slot"]]
            if m.get("codes"):
                assert kind in (None, "pooled_g8")
                g_in = ops.grad_pool_codes_g8(g, y) if g is not None else None      # y: the winner codes
            elif m["g8"]:
                assert kind in (None, "g8")
                g_in = g                                        # gated by the launch above
            else:
                chain = g if kind == "f32" else None
                pooled = g if kind == "pooled" else None
                add_post, add_pre = grads.get(("post", idx)), grads.get(("pre", idx))
                aff_post, aff_pre = affs.get(("post", idx)), affs.get(("pre", idx))
                gate = y if m["relu"] else None
                if pooled is not None and (m["shape"][2] % 2 or m["shape"][3] % 2):
                    chain, pooled = ops.maxpool2x2_bwd(y, pooled, False), None      # odd sizes: separate pool backward
                if aff_post is not None and aff_pre is not None:
                    raise RuntimeError("VGGChainFunction: affine loss gradients on both captures of one layer")
                if aff_pre is not None and m["relu"]:
                    # the saved tensor of a pre-captured layer that continues is its PRE-ReLU output: the tap itself
                    pass
                if all(t is None for t in (chain, pooled, add_post, add_pre, aff_post, aff_pre)):
                    g_in = None
                else:
                    # the tap of a post capture is relu(y) when y is the saved pre-ReLU output of a pre-captured layer
                    aff_x = None
                    if aff_post is not None:
                        aff_x = y if idx not in ctx.pre_idx else torch.relu(y)
                    elif aff_pre is not None:
                        aff_x = y
                    g_in = ops.grad_join_g8(m["shape"], chain, pooled, gate, add_post, add_pre, aff_x, aff_post, aff_pre)
            if g_in is None:                                    # no loss above this layer
                kind, g = None, None
                continue
            conv = net.convs[str(idx)]
            wpt = net._packed.get(idx, conv.weight, True, "bf16")
            below = meta[k - 1] if k > 0 else None
            if below is None:
                dx = ops.conv3x3_dgrad_g8(g_in, wpt, m["cin"], dx_g8=False)
            elif below["kind"] == "pool" and below.get("fused"):
                kind, g = "g8", ops.conv3x3_dgrad_g8(g_in, wpt, m["cin"], dx_g8=True)       # the codes carry the gate
            elif below["kind"] == "conv" and below["g8"]:
                kind, g = "g8", ops.conv3x3_dgrad_g8(g_in, wpt, m["cin"], dx_g8=True, dx_gate=saved[below["slot"]])
            else:
                kind, g = "f32", ops.conv3x3_dgrad_g8(g_in, wpt, m["cin"], dx_g8=False)
        if dx is None:                                          # no output carried a gradient
            dx = torch.zeros(ctx.x_shape, dtype=torch.float32, device=saved[0].device)
        return dx, None, None, None, None
