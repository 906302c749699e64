"""Event timeline (no profiler) of four syevd jobs of different sizes on four streams, issued short-first and long-first."""
import sys, time, torch
sys.path.insert(0, ".")
from style_transfer_amd import eigh
def spd(b, c, n):
    x = torch.randn(b, c, n, device="cuda"); x = x - x.mean(-1, keepdim=True); return x @ x.transpose(-1, -2)
ms = [spd(8, 64, 4096), spd(8, 128, 4096), spd(8, 256, 4096), spd(16, 512, 1024)]
streams = [torch.cuda.Stream() for _ in ms]
def run(order):
    cur = torch.cuda.current_stream()
    t0 = torch.cuda.Event(enable_timing=True); t0.record()
    evs = {}
    h0 = time.perf_counter()
    for i in order:
        s = streams[i]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); eigh.syevd(ms[i]); b.record()
        evs[i] = (a, b, (time.perf_counter() - h0) * 1e3)
    for s in streams:
        cur.wait_stream(s)
    torch.cuda.synchronize()
    return {i: (round(t0.elapsed_time(a), 2), round(t0.elapsed_time(b), 2), round(h, 2)) for i, (a, b, h) in evs.items()}
for order in ([0, 1, 2, 3], [3, 2, 1, 0]):
    run(order); run(order)
    print("order", [ms[i].shape[1] for i in order], "-> C: (gpu start, gpu end, host issued by) ms")
    for i, v in run(order).items():
        print(f"    C={ms[i].shape[1]:4d}: {v}", flush=True)
