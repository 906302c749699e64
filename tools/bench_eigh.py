"""Time torch.linalg.eigh / eigvalsh (rocSOLVER) on the VincentStyleLoss shapes: [B, C, C] for the five style layers."""
import sys
import time

import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")


def spd(b, c, n):
    x = torch.randn(b, c, n, device=dev)
    x = x - x.mean(-1, keepdim=True)
    return x @ x.transpose(-1, -2)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for c, n in ((64, 65536), (128, 16384), (256, 4096), (512, 1024)):
    a = spd(B, c, n)
    print(f"C={c:4d} B={B}: eigh {timeit(lambda: torch.linalg.eigh(a)):8.3f} ms   eigvalsh {timeit(lambda: torch.linalg.eigvalsh(a)):8.3f} ms"
          f"   bmm {timeit(lambda: a @ a):7.3f} ms", flush=True)
a = spd(2 * B, 512, 1024)
print(f"C= 512 B={2*B}: eigh {timeit(lambda: torch.linalg.eigh(a)):8.3f} ms (two layers batched)", flush=True)
# concurrency: the five layer-sized problems on five streams vs back to back
mats = [spd(B, c, n) for c, n in ((64, 65536), (128, 16384), (256, 4096), (512, 1024), (512, 1024))]
streams = [torch.cuda.Stream() for _ in mats]


def serial():
    for m in mats:
        torch.linalg.eigh(m)


def parallel():
    cur = torch.cuda.current_stream()
    for m, s in zip(mats, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            torch.linalg.eigh(m)
    for s in streams:
        cur.wait_stream(s)


print(f"five layers: serial {timeit(serial):8.3f} ms   five streams {timeit(parallel):8.3f} ms", flush=True)
