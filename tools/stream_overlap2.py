"""Which kinds of kernels overlap across HIP streams on this box?  A long single-thread spin kernel
(torch.cuda._sleep) on one stream against (1) another spin, (2) a chain of short kernels, (3) a big conv-like GEMM."""
import time, torch
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
spin = lambda: torch.cuda._sleep(int(5e6))
x = torch.randn(1, 256, 4096, device="cuda")
chain = lambda: [torch.softmax(x, dim=-1) for _ in range(300)]
a = torch.randn(8192, 8192, device="cuda"); b = torch.randn(8192, 8192, device="cuda")
gemm = lambda: [a @ b for _ in range(3)]
def both(f, g):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        f()
    with torch.cuda.stream(sb):
        g()
    cur.wait_stream(sa); cur.wait_stream(sb)
for name, f in (("spin", spin), ("chain of 300 short kernels", chain), ("3 big GEMMs", gemm)):
    print(f"{name}: alone {timeit(f):7.3f} ms", flush=True)
print(f"spin || spin: {timeit(lambda: both(spin, spin)):7.3f} ms", flush=True)
print(f"spin || chain: {timeit(lambda: both(spin, chain)):7.3f} ms", flush=True)
print(f"chain || spin: {timeit(lambda: both(chain, spin)):7.3f} ms", flush=True)
print(f"spin || gemm: {timeit(lambda: both(spin, gemm)):7.3f} ms", flush=True)
print(f"gemm || chain: {timeit(lambda: both(gemm, chain)):7.3f} ms", flush=True)
print(f"chain || gemm: {timeit(lambda: both(chain, gemm)):7.3f} ms", flush=True)
