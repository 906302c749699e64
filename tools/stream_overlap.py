"""Do chains of tiny kernels on different HIP streams overlap on this box?  (a) plain torch elementwise chains,
(b) rocSOLVER syevd pairs of equal size, (c) mixed sizes."""
import sys, time, torch
sys.path.insert(0, ".")
from style_transfer_amd import eigh

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def spd(b, c, n):
    x = torch.randn(b, c, n, device="cuda"); x = x - x.mean(-1, keepdim=True); return x @ x.transpose(-1, -2)

streams = [torch.cuda.Stream() for _ in range(4)]
def on_streams(jobs):
    cur = torch.cuda.current_stream()
    for job, s in zip(jobs, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            job()
    for s in streams[:len(jobs)]:
        cur.wait_stream(s)

# (a) long-ish kernels with small grids: 64 blocks each, ~50 us
xs = [torch.randn(64, 256, 4096, device="cuda") for _ in range(4)]
def chain(x):
    def f():
        for _ in range(100):
            torch.softmax(x[:1], dim=-1)          # 256 rows -> small grid, tens of us
    return f
jobs = [chain(x) for x in xs]
print(f"(a) torch chains x4: serial {timeit(lambda: [j() for j in jobs]):7.3f} ms, 4 streams {timeit(lambda: on_streams(jobs)):7.3f} ms", flush=True)
for c, b in ((512, 8), (256, 8), (64, 8)):
    ms = [spd(b, c, 4096) for _ in range(2)]
    jobs = [(lambda m=m: eigh.syevd(m)) for m in ms]
    print(f"(b) syevd C={c} x2: serial {timeit(lambda: [j() for j in jobs]):7.3f} ms, 2 streams {timeit(lambda: on_streams(jobs)):7.3f} ms", flush=True)
ms = [spd(8, 64, 4096), spd(8, 128, 4096), spd(8, 256, 4096), spd(16, 512, 1024)]
jobs = [(lambda m=m: eigh.syevd(m)) for m in ms]
print(f"(c) syevd 64/128/256/512: serial {timeit(lambda: [j() for j in jobs]):7.3f} ms, 4 streams {timeit(lambda: on_streams(jobs)):7.3f} ms", flush=True)
jobs = jobs[::-1]
print(f"(c') reversed order:       serial {timeit(lambda: [j() for j in jobs]):7.3f} ms, 4 streams {timeit(lambda: on_streams(jobs)):7.3f} ms", flush=True)

# (d) same as (c) but WITHOUT any event dependencies between streams (device-wide sync at the end instead)
jobs = jobs[::-1]
def no_events():
    for job, s in zip(jobs, streams):
        with torch.cuda.stream(s):
            job()
print(f"(d) forward order, no cross-stream events: 4 streams {timeit(no_events):7.3f} ms", flush=True)
# (e) with a non-null main stream
main = torch.cuda.Stream()
with torch.cuda.stream(main):
    print(f"(e) forward order, non-null main stream: serial {timeit(lambda: [j() for j in jobs]):7.3f} ms, 4 streams {timeit(lambda: on_streams(jobs)):7.3f} ms", flush=True)
