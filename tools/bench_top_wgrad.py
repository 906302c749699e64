#!/usr/bin/env python3
"""Weight gradient of the decoder's 64 -> 3 top layer from a G8 input (st_conv3x3_wgrad_cout3_xg8) at B=8, 512 x 512
and of the fp32 form (st_conv3x3_wgrad, cout = 3) at B=8, 256 x 256."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from style_transfer_amd import ops  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return a.elapsed_time(e) / iters * 1e3


def main():
    x = torch.relu(torch.randn(8, 64, 512, 512, device="cuda"))
    dy = torch.randn(8, 3, 512, 512, device="cuda")
    xg = ops.to_g8(x)
    us = timed(lambda: ops.conv3x3_wgrad_cout3_g8(xg, dy))
    print(f"wgrad 64->3 G8 x   B=8 512^2  {us:7.1f} us")
    x2 = torch.relu(torch.randn(8, 64, 256, 256, device="cuda"))
    dy2 = torch.randn(8, 3, 256, 256, device="cuda")
    us = timed(lambda: ops.conv3x3_wgrad(x2, dy2, None))
    print(f"wgrad 64->3 fp32 x B=8 256^2  {us:7.1f} us")


if __name__ == "__main__":
    main()
