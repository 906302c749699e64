#!/usr/bin/env python3
"""The staged bf16 weight gradient inside the decoder's G8 chain (st_conv3x3_wgrad_bf16_pre: x in G8, g16 ready) on
the decoder's layers at batch 8, 512 x 512 output; per-kernel times come from rocprofv3 (tools/prof_cmd.sh), the
printed figure is the whole call (prep + GEMM + reduces) on HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from style_transfer_amd import ops

# (name, cin, cout, input side, upsample)
LAYERS = [("dec L0", 512, 256, 64, False), ("dec L1 up", 256, 256, 64, True), ("dec L2/3", 256, 256, 128, False),
          ("dec L4", 256, 128, 128, False), ("dec L5 up", 128, 128, 128, True), ("dec L6", 128, 64, 256, False),
          ("dec L7 up", 64, 64, 256, True)]


def main():
    n = int(os.environ.get("B", "8"))
    only = os.environ.get("ONLY")
    tot = totf = 0.0
    for name, cin, cout, side, up in LAYERS:
        if only and only not in name:
            continue
        so = side * (2 if up else 1)
        x = ops.to_g8(torch.randn(n, cin, side, side, device="cuda"))
        g16 = torch.randn(n, cout, so, so, device="cuda").to(torch.bfloat16)
        slabs = torch.randn(64, cout, device="cuda")
        fn = lambda: ops.conv3x3_wgrad_pre(x, g16, slabs, up, ops.PAD_REPLICATE)
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        fl = 2.0 * n * so * so * cin * cout * 9
        tot += ms
        totf += fl
        print(f"{name:10s} {cin:4d}->{cout:4d} out {so:4d}^2  {ms * 1e3:8.1f} us  {fl / ms / 1e9:8.1f} TF/s", flush=True)
    print(f"sum {tot:.3f} ms  {totf / tot / 1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
