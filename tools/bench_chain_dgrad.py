#!/usr/bin/env python3
"""Per-layer timing of the decoder chain's bf16 data gradient (st_conv3x3_dgrad_bf16_g8_chain) at the BASELINE
configs[3] per-GPU shape (batch 8, 512x512): fused epilogue + band kernels vs. padded workspace + fold (ST_NO_FUSED_FOLD=1).
    python tools/bench_chain_dgrad.py [--batch 8] [--side 512]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--side", type=int, default=512)
args = ap.parse_args()
n, S = args.batch, args.side
# (name, k = layer cout, c = layer cin, conv-input side, upsample, dy form, gate form)
LAYERS = [("L8 64<-3", 3, 64, S, False, "f32", "f32"), ("L7 64<-64 up", 64, 64, S, True, "g8", "g8"),
          ("L6 128<-64", 64, 128, S // 2, False, "g8", "g8"), ("L5 128<-128 up", 128, 128, S // 2, True, "g8", "g8"),
          ("L4 256<-128", 128, 256, S // 4, False, "g8", "g8"), ("L3 256<-256", 256, 256, S // 4, False, "g8", "g8"),
          ("L1 256<-256 up", 256, 256, S // 4, True, "g8", "g8"), ("L0 512<-256 f32out", 256, 512, S // 8, False, "g8", None)]
g = torch.Generator(device="cuda").manual_seed(0)
tot = {"fused": 0.0, "workspace": 0.0}
for name, k, c, side, up, dyf, gf in LAYERS:
    wt = torch.randn(k, c, 3, 3, device="cuda", generator=g) * 0.05
    wpt = ops.pack_conv3x3_weights(wt, transpose_flip=True, compute="bf16")
    dy = torch.randn(n, k, side, side, device="cuda", generator=g)
    dy = ops.to_g8(dy) if dyf == "g8" else dy
    s = 2 if up else 1
    act = torch.relu(torch.randn(n, c, side // s, side // s, device="cuda", generator=g))
    gate = ops.to_g8(act) if gf == "g8" else (act if gf == "f32" else None)
    bottom = gf is None
    row = []
    for mode in ("fused", "workspace"):
        if mode == "workspace":
            os.environ["ST_NO_FUSED_FOLD"] = "1"
        else:
            os.environ.pop("ST_NO_FUSED_FOLD", None)
        f = (lambda: ops.conv3x3_dgrad_g8_chain(dy, wpt, c, up, None, True, False, False, False)) if bottom else \
            (lambda: ops.conv3x3_dgrad_g8_chain(dy, wpt, c, up, gate, False, True, True, True))
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 100
        tot[mode] += us
        row.append(us)
    os.environ.pop("ST_NO_FUSED_FOLD", None)
    gf_ = 2.0 * n * side * side * k * c * 9 / 1e9
    print(f"{name:18s} {gf_:7.1f} GF  fused {row[0]:8.1f} us ({gf_ / row[0] * 1e3:7.1f} TF/s)   workspace+fold {row[1]:8.1f} us")
print(f"sum: fused {tot['fused']:.0f} us, workspace+fold {tot['workspace']:.0f} us")
