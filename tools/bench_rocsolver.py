"""Direct rocSOLVER calls (ctypes) for the batched symmetric eigenproblems of VincentStyleLoss: which driver is
fastest at [B, C, C], and does skipping torch.linalg's per-call `info` read let calls overlap across streams?"""
import ctypes
import sys
import time

import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
rb = ctypes.CDLL("librocblas.so")
rs = ctypes.CDLL("librocsolver.so")
P, I, L, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float
EV_ORIG, EV_NONE, LOWER, SORT_ASC = 211, 213, 122, 252


def handle(stream):
    h = P()
    assert rb.rocblas_create_handle(ctypes.byref(h)) == 0
    rb.rocblas_set_stream.argtypes = [P, P]
    assert rb.rocblas_set_stream(h, P(stream)) == 0
    return h


rs.rocsolver_ssyevd_strided_batched.argtypes = [P, I, I, I, P, I, L, P, L, P, L, P, I]
rs.rocsolver_ssyev_strided_batched.argtypes = [P, I, I, I, P, I, L, P, L, P, L, P, I]
rs.rocsolver_ssyevdj_strided_batched.argtypes = [P, I, I, I, P, I, L, P, L, P, I]
rs.rocsolver_ssyevj_strided_batched.argtypes = [P, I, I, I, I, P, I, L, F, P, I, P, P, L, P, I]


def spd(b, c, n):
    x = torch.randn(b, c, n, device=dev)
    x = x - x.mean(-1, keepdim=True)
    return x @ x.transpose(-1, -2)


def run(h, name, a):
    b, c, _ = a.shape
    work = a.clone()
    w = torch.empty(b, c, device=dev)
    e = torch.empty(b, c, device=dev)
    info = torch.zeros(b, dtype=torch.int32, device=dev)
    if name in ("syevd", "syev"):
        fn = getattr(rs, f"rocsolver_s{name}_strided_batched")
        rc = fn(h, EV_ORIG, LOWER, c, work.data_ptr(), c, c * c, w.data_ptr(), c, e.data_ptr(), c, info.data_ptr(), b)
    elif name == "syevdj":
        rc = rs.rocsolver_ssyevdj_strided_batched(h, EV_ORIG, LOWER, c, work.data_ptr(), c, c * c, w.data_ptr(), c,
                                                  info.data_ptr(), b)
    else:
        res = torch.empty(b, device=dev)
        nsw = torch.zeros(b, dtype=torch.int32, device=dev)
        rc = rs.rocsolver_ssyevj_strided_batched(h, SORT_ASC, EV_ORIG, LOWER, c, work.data_ptr(), c, c * c, 0.0,
                                                 res.data_ptr(), 100, nsw.data_ptr(), w.data_ptr(), c, info.data_ptr(), b)
    assert rc == 0, (name, rc)
    return w, work, info


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


h = handle(torch.cuda.current_stream().cuda_stream)
for c, n, b in ((64, 65536, B), (128, 16384, B), (256, 4096, B), (512, 1024, B), (512, 1024, 2 * B)):
    a = spd(b, c, n)
    ref_w, ref_v = torch.linalg.eigh(a)
    line = f"C={c:4d} B={b:2d}: torch.eigh {timeit(lambda: torch.linalg.eigh(a)):7.3f} ms"
    for name in ("syevd", "syev", "syevdj", "syevj"):
        try:
            w, v, info = run(h, name, a)
            torch.cuda.synchronize()
            err = float((w - ref_w).abs().max() / ref_w.abs().max())
            # row-major view of the column-major output: rows are eigenvectors -> V = v^T
            V = v.transpose(-1, -2)
            rec = float((V @ torch.diag_embed(w) @ V.transpose(-1, -2) - a).abs().max() / a.abs().max())
            line += f" | {name} {timeit(lambda: run(h, name, a)):7.3f} ms (w err {err:.1e}, recon {rec:.1e}, info {int(info.abs().max())})"
        except Exception as ex:   # noqa: BLE001
            line += f" | {name} failed: {ex}"
    print(line, flush=True)

# overlap across streams without host syncs: five layer-sized problems
mats = [spd(B, c, n) for c, n in ((64, 65536), (128, 16384), (256, 4096), (512, 1024), (512, 1024))]
streams = [torch.cuda.Stream() for _ in mats]
handles = [handle(s.cuda_stream) for s in streams]


def serial():
    for m in mats:
        run(h, "syevd", m)


def parallel():
    cur = torch.cuda.current_stream()
    for m, s, hh in zip(mats, streams, handles):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            run(hh, "syevd", m)
    for s in streams:
        cur.wait_stream(s)


print(f"five layers, syevd direct: serial {timeit(serial):7.3f} ms   five streams {timeit(parallel):7.3f} ms", flush=True)
