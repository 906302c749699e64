#!/usr/bin/env python3
"""Back-to-back timing of st_adain_fwd on the relu4_1 feature shape of BASELINE configs[1]
(16 x 512 planes of 64 x 64): two alternating feature sets (805 MB working set, no Infinity Cache help)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from style_transfer_amd import ops  # noqa: E402


def run(n, c, hw, reps=20):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    sets = [(torch.rand(n, c, hw, hw, device=dev, generator=g), torch.rand(n, c, hw, hw, device=dev, generator=g))
            for _ in range(2)]
    for fc, fs in sets:
        ops.adain(fc, fs, 1e-5, 1e-5, 1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for fc, fs in sets:
            ops.adain(fc, fs, 1e-5, 1e-5, 1.0)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (2 * reps)
    byt = 3 * 4 * n * c * hw * hw
    return us, byt / us / 1e3


if __name__ == "__main__":
    for pb in [None]:
        for shape in ((16, 512, 64), (8, 512, 32), (1, 512, 256), (8, 512, 128)):
            us, gbs = run(*shape)
            print(f"shape={shape}: {us:.2f} us  {gbs:.0f} GB/s  {gbs / 80:.1f}% of 8 TB/s", flush=True)
