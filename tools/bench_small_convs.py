#!/usr/bin/env python3
"""The two 3-channel layers (conv1_1 3->64, decoder's last conv 64->3) against HBM: time and GB/s at the shapes of
BASELINE configs[1] (B=16, 512^2), configs[4] (1 x 2048^2) and configs[2] (B=8, 256^2)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from style_transfer_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda")
    for b, s in ((16, 512), (1, 2048), (8, 256)):
        for cin, cout in ((3, 64), (64, 3)):
            x = torch.randn(b, cin, s, s, device=dev)
            w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
            bias = torch.randn(cout, device=dev)
            wp = ops.pack_conv3x3_weights(w)
            fn = lambda: ops.conv3x3(x, wp, bias, cout)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 50
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) / iters * 1e3
            nbytes = 4.0 * b * s * s * (cin + cout)
            print(f"conv3x3 {cin:2d}->{cout:2d}  B={b:2d} {s}^2  {us:8.1f} us  {nbytes / us / 1e3:6.0f} GB/s  "
                  f"{nbytes / us / 1e3 / 80:5.1f}% of HBM peak", flush=True)


if __name__ == "__main__":
    main()
