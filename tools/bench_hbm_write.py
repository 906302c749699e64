#!/usr/bin/env python3
"""What a write-only stream reaches on this GPU (the 3 -> 64 first layer writes 21 bytes for every byte it reads):
torch's fill and copy kernels on 1 GiB, HIP events."""
import torch


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    n = 1 << 28
    x = torch.empty(n, device="cuda")
    y = torch.empty(n, device="cuda")
    us = timed(lambda: x.fill_(1.0))
    print(f"fill  1 GiB (write only)      {us:8.1f} us  {4.0 * n / us / 1e3:6.0f} GB/s written")
    us = timed(lambda: y.copy_(x))
    print(f"copy  1 GiB (read + write)    {us:8.1f} us  {8.0 * n / us / 1e3:6.0f} GB/s moved")
    us = timed(lambda: x.sum())
    print(f"sum   1 GiB (read only)       {us:8.1f} us  {4.0 * n / us / 1e3:6.0f} GB/s read")


if __name__ == "__main__":
    main()
