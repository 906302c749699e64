#!/usr/bin/env python3
"""G8 -> G8 launches of the loss VGG at batch 8, 512 x 512: forward vs data gradient without / with the output gate
(st_conv3x3_dgrad_bf16_g8's ogate).  Same FLOPs per row; HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from style_transfer_amd import ops

LAYERS = [("conv1_2", 64, 64, 512), ("conv2_2", 128, 128, 256), ("conv3_2", 256, 256, 128), ("conv4_1", 256, 512, 64)]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    n = 8
    for name, cin, cout, side in LAYERS:
        x = ops.to_g8(torch.randn(n, cin, side, side, device="cuda"))
        dy = ops.to_g8(torch.randn(n, cout, side, side, device="cuda"))
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        b = torch.randn(cout, device="cuda")
        wp = ops.pack_conv3x3_weights(w, compute="bf16")
        wpt = ops.pack_conv3x3_weights(w, transpose_flip=True, compute="bf16")
        fl = 2.0 * n * side * side * cin * cout * 9
        rows = [("fwd g8->g8", lambda: ops.conv3x3_g8(x, wp, b, cout, pad_mode=ops.PAD_ZERO, y_g8=True)),
                ("dgrad g8->g8", lambda: ops.conv3x3_dgrad_g8(dy, wpt, cin, dx_g8=True)),
                ("dgrad g8->g8 gated", lambda: ops.conv3x3_dgrad_g8(dy, wpt, cin, dx_g8=True, dx_gate=x)),
                ("dgrad g8->f32", lambda: ops.conv3x3_dgrad_g8(dy, wpt, cin, dx_g8=False))]
        for what, fn in rows:
            ms = timed(fn)
            print(f"{name:8s} {cin:4d}->{cout:4d} {side:4d}^2  {what:20s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:8.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
