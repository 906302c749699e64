#!/usr/bin/env python3
"""Timing of the non-conv kernels of the training step against their rooflines:
RGB-uv histogram block, TV loss, mean/std statistics, MSE, Normalize (HBM bytes) and the covariance
("Gram") kernel (MFMA FLOPs), forward and backward through the module API, back-to-back launches
between one event pair.  Prints one line per case."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from style_transfer_amd import ops  # noqa: E402
from style_transfer_amd.losses import AdaINContentLoss, AdaINStyleLoss, HistLoss, TVLoss, VincentStyleLoss  # noqa: E402

HBM, FP32_MFMA = 8000.0, 157.3


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


NO_GRAPH = "--no-graph" in sys.argv      # counter passes: plain launches only


def graphed(fn):
    """fn captured in a HIP graph: the replay time is the GPU time of its kernels without the host's launch and
    autograd overhead (what a training step replayed from graphs pays).  Falls back to fn if capture fails."""
    if NO_GRAPH:
        return fn
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g.replay
    except Exception as e:       # noqa: BLE001
        print(f"   (graph capture failed: {type(e).__name__}: {e}; eager timing)", flush=True)
        torch.cuda.synchronize()
        return fn


def line(name, us, nbytes=None, flops=None):
    s = f"{name:58s} {us:9.1f} us"
    if nbytes:
        gbs = nbytes / us / 1e3
        s += f"  {nbytes / 1e6:8.2f} MB  {gbs:7.0f} GB/s  {100 * gbs / HBM:5.1f}% of HBM peak"
    if flops:
        tf = flops / us / 1e6
        s += f"  {flops / 1e9:8.2f} GF  {tf:6.1f} TF/s  {100 * tf / FP32_MFMA:5.1f}% of fp32 MFMA peak"
    print(s, flush=True)


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hist, tv, sty, mse, vin = HistLoss(), TVLoss(), AdaINStyleLoss(), AdaINContentLoss(), VincentStyleLoss()
    for b, s in ((8, 256), (8, 512), (16, 2048)):
        x = torch.rand(b, 3, s, s, device=dev, generator=g)
        with torch.no_grad():
            us = timeit(lambda: hist.histogram_block(x))
        line(f"rgbuv_hist fwd  B={b} S={s}", us, nbytes=4.0 * (x.numel() + b * 3 * 64 * 64))
        xg = x.clone().requires_grad_(True)
        t = torch.rand(b, 3, 64, 64, device=dev, generator=g)

        def fb():
            xg.grad = None
            (hist.histogram_block(xg) * t).sum().backward()
        line(f"rgbuv_hist fwd+bwd  B={b} S={s}", timeit(fb), nbytes=4.0 * (3 * x.numel()))
        line(f"rgbuv_hist fwd+bwd  B={b} S={s}  (graph replay)", timeit(graphed(fb)), nbytes=4.0 * (3 * x.numel()))
        with torch.no_grad():
            line(f"tv fwd  B={b} S={s}", timeit(lambda: tv(x)), nbytes=4.0 * x.numel())
            line(f"normalize fwd  B={b} S={s}", timeit(lambda: ops.normalize(x, [.485, .456, .406], [.229, .224, .225])),
                 nbytes=8.0 * x.numel())
    for b, c, hw in ((8, 64, 256), (8, 128, 128), (8, 256, 64), (8, 512, 32), (16, 512, 64)):
        f = torch.randn(b, c, hw, hw, device=dev, generator=g)
        f2 = torch.randn(b, c, hw, hw, device=dev, generator=g)
        with torch.no_grad():
            line(f"mean_std  [{b},{c},{hw},{hw}]", timeit(lambda: ops.mean_std(f)), nbytes=4.0 * f.numel())
            line(f"mse fwd  [{b},{c},{hw},{hw}]", timeit(lambda: mse(f, f2)), nbytes=8.0 * f.numel())
            line(f"covariance (Gram) fwd  [{b},{c},{hw},{hw}]", timeit(lambda: vin.calc_2_moments(f)),
                 flops=2.0 * b * c * c * hw * hw)
        fc = f.clone().requires_grad_(True)
        gc = torch.randn(b, c, c, device=dev, generator=g)

        def cb():
            fc.grad = None
            mu, cov = vin.calc_2_moments(fc)
            (cov * gc).sum().backward()
        line(f"covariance (Gram) fwd+bwd  [{b},{c},{hw},{hw}]", timeit(cb), flops=6.0 * b * c * c * hw * hw)
        line(f"covariance (Gram) fwd+bwd  [{b},{c},{hw},{hw}]  (graph replay)", timeit(graphed(cb)),
             flops=6.0 * b * c * c * hw * hw)
        fg = f.clone().requires_grad_(True)

        def sb():
            fg.grad = None
            sty(fg, f2).backward()
        line(f"mean/std style loss fwd+bwd  [{b},{c},{hw},{hw}]", timeit(sb), nbytes=4.0 * 4 * f.numel())
        line(f"mean/std style loss fwd+bwd  [{b},{c},{hw},{hw}]  (graph replay)", timeit(graphed(sb)),
             nbytes=4.0 * 4 * f.numel())

    # ---- the G8 glue kernels of the bf16 training chains (configs[3] per-GPU shapes: B=8, 512x512 image) ------------
    # algorithmic bytes: every source element read once (fp32 = 4 B), every result written once (G8 / bf16 = 2 B)
    for c, hw in ((64, 512), (128, 256), (256, 128)):
        b = 8
        x = torch.relu(torch.randn(b, c, hw, hw, device=dev, generator=g))
        post = torch.randn(b, c, hw, hw, device=dev, generator=g)
        pooled = torch.randn(b, c, hw // 2, hw // 2, device=dev, generator=g)
        n = x.numel()
        with torch.no_grad():
            line(f"maxpool2x2 -> G8  [{b},{c},{hw},{hw}]", timeit(lambda: ops.maxpool2x2_g8(x)), nbytes=4.0 * n + 0.5 * n)
            line(f"grad join through the pool (x, tap grad, pooled grad -> G8)  [{b},{c},{hw},{hw}]",
                 timeit(lambda: ops.grad_join_g8(x.shape, chain_pooled=pooled, x_gate=x, add_post=post)),
                 nbytes=4.0 * n * 2 + 1.0 * n + 2.0 * n)
            line(f"grad join (chain + tap grad, gate -> G8)  [{b},{c},{hw},{hw}]",
                 timeit(lambda: ops.grad_join_g8(x.shape, chain=post, x_gate=x, add_post=post)),
                 nbytes=4.0 * n * 3 + 2.0 * n)
    for cin, hw, up in ((64, 512, False), (64, 512, True), (128, 256, False), (256, 128, True)):
        b, cout = 8, cin
        wt = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * (2.0 / (9 * cin)) ** 0.5
        wpt = ops.pack_conv3x3_weights(wt, True, "bf16")
        dy = ops.to_g8(torch.randn(b, cout, hw, hw, device=dev, generator=g))
        s_ = 2 if up else 1
        gate = ops.to_g8(torch.relu(torch.randn(b, cin, hw // s_, hw // s_, device=dev, generator=g)))

        def dg():
            return ops.conv3x3_dgrad_g8_chain(dy, wpt, cin, up, gate, want_f32=False, want_g8=True, want_g16=True,
                                              want_bias=True)
        with torch.no_grad():
            t_all = timeit(dg)
        # the fold alone = the call minus its GEMM (the same GEMM writing fp32 through the forward entry point)
        wp_f = ops.pack_conv3x3_weights(wt, False, "bf16")
        with torch.no_grad():
            t_gemm = timeit(lambda: ops.conv3x3_g8(dy, wp_f, None, cin, False, ops.PAD_ZERO, relu=False, y_g8=False))
        n_pad = b * cin * (hw + 2) * ((hw + 2 + 3) // 4 * 4)
        n_src = b * cin * (hw // s_) ** 2
        line(f"dgrad G8 chain call, {cout}->{cin} at {hw}x{hw}{' (x2 upsample below)' if up else ''}: GEMM + fold8",
             t_all, flops=2.0 * b * hw * hw * cin * cout * 9)
        line("   of which dgrad_fold8 (padded fp32 + G8 gate -> G8 + bf16 G + bias sums), by difference",
             max(t_all - t_gemm, 0.1), nbytes=4.0 * n_pad + 2.0 * n_src + 2.0 * n_src * 2)


if __name__ == "__main__":
    main()
