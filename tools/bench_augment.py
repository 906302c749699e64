#!/usr/bin/env python3
"""Post-ToTensor augmentation of one training batch pair (content + style, src/train.py:116-135): the HIP kernels
of csrc/augment.hip on a batch in HBM against the same transforms per image on the host (transforms.py, one core --
what one DataLoader worker does).  Prints the GPU time per batch pair, its HBM rate, and the CPU time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from style_transfer_amd import transforms as T  # noqa: E402
from style_transfer_amd.gpu_augment import GpuAugment  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    for n, s in ((8, 256), (64, 256), (16, 512)):
        g = torch.Generator().manual_seed(0)
        xc, xs = torch.rand(n, 3, s, s, generator=g), torch.rand(n, 3, s, s, generator=g)
        dc, ds = xc.to(dev), xs.to(dev)
        ac, as_ = GpuAugment.for_content(seed=0), GpuAugment.for_style(seed=1)
        for _ in range(3):
            ac(dc), as_(ds)
        torch.cuda.synchronize()
        iters = 30
        t0 = time.perf_counter()
        for _ in range(iters):
            ac(dc), as_(ds)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / iters * 1e3
        # device time alone: explicit parameters, no host draws inside the event pair
        pc, ps = ac.draw(n), as_.draw(n)
        nf = torch.randn(n, 3, s, s, device=dev) * 0.05 + 0.5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ac(dc, params=pc), as_(ds, params=ps, noise_field=nf)
        e1.record()
        e1.synchronize()
        gpu = e0.elapsed_time(e1) / iters
        # passes over an image: content rot+sharp+4 jitter+flip = 7 (read+write), style adds noise (2 rw + 2 reads + noise read)
        nbytes = 4.0 * xc.numel() * (7 * 2 + 7 * 2 + 2 * 2 + 3)
        msg = (f"augment pair  n={n} {s}x{s}:  GPU {gpu:7.3f} ms (events, explicit draws)  {wall:7.3f} ms wall with host draws"
               f"  ~{nbytes / gpu / 1e6:6.0f} GB/s over {nbytes / 1e6:.0f} MB")
        if n <= 16:
            content = T.Compose([T.RandomRotation(90), T.RandomAdjustSharpness(1.5, 0.5),
                                 T.ColorJitter(0.2, 0.4, 0.3, 0.1), T.RandomHorizontalFlip(0.5)])
            style = T.Compose([T.RandomRotation(90), T.AddGaussianNoise(0.5, (0., 0.08), 0.5), T.RandomAdjustSharpness(1.5, 0.5),
                               T.ColorJitter(0.2, 0.25, 0.3, 0.1), T.RandomHorizontalFlip(0.5)])
            torch.set_num_threads(1)
            t0 = time.perf_counter()
            for i in range(n):
                content(xc[i]), style(xs[i])
            cpu = (time.perf_counter() - t0) * 1e3
            msg += f"   CPU (1 core, per image) {cpu:8.1f} ms  = {cpu / gpu:6.0f}x"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
