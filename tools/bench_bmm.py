#!/usr/bin/env python3
"""st_bmm_f32 (csrc/bmm.hip) against torch.bmm (rocBLAS) on the C x C shapes of the Gram loss; HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from style_transfer_amd import ops


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for batch, n in [(16, 512), (8, 512), (8, 256), (16, 256), (8, 128), (8, 64)]:
    a = torch.randn(batch, n, n, device="cuda")
    b = torch.randn(batch, n, n, device="cuda")
    fl = 2.0 * batch * n ** 3
    for name, fn in [("st_bmm_f32 NN", lambda: ops.bmm_raw(a, b)), ("st_bmm_f32 NT", lambda: ops.bmm_raw(a, b, False, True)),
                     ("st_bmm_f32 TN", lambda: ops.bmm_raw(a, b, True, False)), ("torch.bmm    NN", lambda: torch.bmm(a, b)),
                     ("torch.bmm    NT", lambda: torch.bmm(a, b.transpose(1, 2)))]:
        us = timed(fn)
        print(f"[{batch:2d},{n:3d},{n:3d}] {name}  {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s", flush=True)
