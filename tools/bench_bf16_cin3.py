#!/usr/bin/env python3
"""The bf16-mode 3 -> 64 first layer (conv1_1 of the encoder / the loss VGG) at B=8 and 16, 512 x 512: fp32 NCHW and G8
results; ST_BF16_CIN3_TAIL=1 selects the general kernel's TAIL form."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from style_transfer_amd import ops  # noqa: E402


def main():
    for b in (8, 16):
        x = torch.rand(b, 3, 512, 512, device="cuda")
        w = torch.randn(64, 3, 3, 3, device="cuda") * 0.1
        bias = torch.randn(64, device="cuda") * 0.1
        wp = ops.pack_conv3x3_weights(w, False, "bf16")
        for g8 in (True, False):
            fn = lambda: ops.conv3x3_g8(x, wp, bias, 64, False, ops.PAD_ZERO, relu=True, pool=False, y_g8=g8)
            for _ in range(3):
                fn()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30):
                fn()
            e.record()
            e.synchronize()
            us = a.elapsed_time(e) / 30 * 1e3
            nbytes = b * 512 * 512 * (3 * 4 + 64 * (2 if g8 else 4))
            print(f"3->64 B={b:2d} 512^2 -> {'G8  ' if g8 else 'fp32'}  {us:7.1f} us  {nbytes / us / 1e3:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
