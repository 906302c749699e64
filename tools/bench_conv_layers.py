#!/usr/bin/env python3
"""Per-layer timing of the fp32 / bf16 MFMA conv3x3 kernels on the encoder/decoder shapes
(BASELINE configs[1]: batch 16 at 512x512).  HIP events on the launch stream.

    python tools/bench_conv_layers.py [--batch 16] [--side 512] [--iters 5] [--mode fwd|dgrad|wgrad] [--compute f32|bf16]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from style_transfer_amd import ops

# (name, cin, cout, side_divisor_of_input, upsample)
LAYERS = [
    ("enc conv1_1", 3, 64, 1, False), ("enc conv1_2", 64, 64, 1, False),
    ("enc conv2_1", 64, 128, 2, False), ("enc conv2_2", 128, 128, 2, False),
    ("enc conv3_1", 128, 256, 4, False), ("enc conv3_x", 256, 256, 4, False),
    ("enc conv4_1", 256, 512, 8, False),
    ("dec 1", 512, 256, 8, False), ("dec 5 up", 256, 256, 8, True), ("dec 8/11", 256, 256, 4, False),
    ("dec 14", 256, 128, 4, False), ("dec 18 up", 128, 128, 4, True), ("dec 21", 128, 64, 2, False),
    ("dec 25 up", 64, 64, 2, True), ("dec 28", 64, 3, 1, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--mode", default="fwd", choices=["fwd", "dgrad", "dgrad_nogate", "wgrad"])
    ap.add_argument("--pad", default="replicate", choices=["replicate", "zero"])
    ap.add_argument("--compute", default="f32", choices=["f32", "bf16", "bf16x3", "bf16g8"],
                    help="bf16g8: bf16 kernels with the G8 bf16 activation layout on both sides where the layer allows")
    args = ap.parse_args()
    dev = torch.device("cuda")
    tot_t = tot_f = 0.0
    for name, cin, cout, div, up in LAYERS:
        h = args.side // div
        x = torch.randn(args.batch, cin, h, h, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        if args.compute == "bf16g8":
            assert args.mode == "fwd"
            wp = ops.pack_conv3x3_weights(w, compute="bf16")
            xin = ops.to_g8(x) if cin % 16 == 0 else x
            yg = cout % 8 == 0
            fn = lambda: ops.conv3x3_g8(xin, wp, b, cout, upsample2x=up, y_g8=yg)
        else:
            wp = ops.pack_conv3x3_weights(w, compute=args.compute)
            y = ops.conv3x3(x, wp, b, cout, upsample2x=up)
            dy = torch.randn_like(y)
            wpt = ops.pack_conv3x3_weights(w, transpose_flip=True, compute=args.compute)
        if args.compute == "bf16g8":
            pass
        elif args.mode == "fwd":
            fn = lambda: ops.conv3x3(x, wp, b, cout, upsample2x=up)
        elif args.mode == "dgrad":
            fn = lambda: ops.conv3x3_dgrad(dy, y, wpt, cin, up and args.pad == "replicate", args.pad)
        elif args.mode == "dgrad_nogate":
            fn = lambda: ops.conv3x3_dgrad(dy, None, wpt, cin, up and args.pad == "replicate", args.pad)
        else:
            fn = lambda: ops.conv3x3_wgrad(x, dy, y, True, up, "replicate", compute=args.compute)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        ho = h * (2 if up else 1)
        fl = 2.0 * args.batch * ho * ho * cin * cout * 9
        tot_t += ms
        tot_f += fl
        print(f"{name:14s} cin={cin:3d} cout={cout:3d} out={ho:4d}^2  {ms:8.3f} ms  {fl / ms / 1e9:7.2f} TF/s", flush=True)
    print(f"sum {tot_t:.2f} ms, {tot_f / tot_t / 1e9:.2f} TF/s over the listed layers")


if __name__ == "__main__":
    main()
