"""Host enqueue cost vs GPU time of rocSOLVER's syevd at the VincentStyleLoss shapes (is a step's eigen work bound by
the host issuing thousands of tiny launches?), and whether several Python threads can issue calls concurrently
(ctypes releases the GIL).  A HIP-graph capture of syevd is refused by rocSOLVER (status 6, internal error)."""
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from style_transfer_amd import eigh  # noqa: E402

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


shapes = ((64, 65536, 8), (128, 16384, 8), (256, 4096, 8), (512, 1024, 8), (512, 1024, 16))
mats = []
for c, n, b in shapes:
    a = spd(b, c, n)
    mats.append(a)
    eigh.syevd(a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    eigh.syevd(a)
    host = (time.perf_counter() - t0) * 1e3                # enqueue time only (no sync)
    torch.cuda.synchronize()
    print(f"C={c:4d} B={b:2d}: eager {timeit(lambda: eigh.syevd(a)):7.3f} ms, of which host enqueue {host:6.3f} ms", flush=True)

# four groups of one training step (64, 128, 256, 512 x 2 layers) issued from four host threads on four streams
group = [mats[0], mats[1], mats[2], mats[4]]
streams = [torch.cuda.Stream() for _ in group]


def one(m, s):
    with torch.cuda.stream(s):
        eigh.syevd(m)


def serial_streams():
    cur = torch.cuda.current_stream()
    for m, s in zip(group, streams):
        s.wait_stream(cur)
        one(m, s)
    for s in streams:
        cur.wait_stream(s)


def threaded():
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    ts = [threading.Thread(target=one, args=(m, s)) for m, s in zip(group, streams)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for s in streams:
        cur.wait_stream(s)


print(f"four groups: one stream {timeit(lambda: [eigh.syevd(m) for m in group]):7.3f} ms | four streams, one host thread "
      f"{timeit(serial_streams):7.3f} ms | four streams, four host threads {timeit(threaded):7.3f} ms", flush=True)
