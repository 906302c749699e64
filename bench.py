#!/usr/bin/env python3
"""Headline benchmark of the AdaIN style-transfer hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode both|infer|train]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

BASELINE.json's metric has two halves and rank 0 prints ONE JSON line carrying both:

  * top-level fields  -- "stylized 512x512 img/s (inference)": BASELINE configs[1], batch 16 at
    512x512 fp32 per GPU (2x encoder + AdaIN + decoder, alpha = 1).  For N > 1 every rank stylises
    its own batch of 16 (images are independent: no data-path collective, weak scaling).
    EXACTLY --steps timed steps after --warmup, barrier + synchronize on both sides, max over ranks.
  * "train" object    -- "train img/s": BASELINE configs[2], batch 8 at 256x256 fp32 per GPU, the full
    training step (2x encoder, AdaIN, decoder, Normalize, loss VGG19, content MSE + mean/std style +
    TV + RGB-uv histogram losses, backward, RCCL bucket all-reduce when N > 1, Adam step), timed the
    same way with its own steps; "gram" inside it is the same step with --use_VincentStyleLoss
    (covariance "Gram" MFMA kernel + eigh).

Extra objects (task brief, section 4):
  roofline     -- the dominant kernel (fp32 MFMA conv3x3, 128-channel tile instantiation): algorithmic
                  FLOPs per launch / average launch duration, measured live with a HIP event pair around
                  every launch on the launch stream, over a repeat of the K timed steps run right after the
                  timed region (inside it the event packets open launch gaps on some boxes: 284-305 vs a
                  steady 308 img/s, same kernel durations); `traffic` = HBM bytes per launch from the
                  committed PMC passes (profiles/, FETCH_SIZE + WRITE_SIZE), not collected live.
  roofline_adain -- the HBM-bound AdaIN kernel: algorithmic bytes (read content, read style, write
                  out) / average launch duration against the 8 TB/s HBM3E peak; 40 launches back to back
                  between one event pair on two alternating feature sets (nothing cached), the in-step
                  per-launch figure next to it.
  cpu_baseline -- the CPU oracle (oracle/ref_ops.py, a port of the reference's torch path) timed on
                  this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 FLOP/clk/CU x 256 CU x 2.4 GHz
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
PREWARM_SECONDS = float(os.environ.get("ST_BENCH_PREWARM", "1.0"))
BATCH, SIDE = 16, 512           # configs[1]
TRAIN_BATCH, TRAIN_SIDE = 8, 256  # configs[2]
CONTENT_IDX, STYLE_IDX = [21], [2, 7, 14, 19, 25]   # CLI defaults, train.py:80-83
DOMINANT = "conv3x3_f32_kernel<2,2,2,2>"
DOMINANT_BF16 = "conv3x3_bf16_kernel<2,4,2,4>"
DOMINANT_BF16X3 = "conv3x3_bf16x3_kernel<1,4,2,4>"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16


def pmc_traffic(kernel: str, fetch_factor: float = 1.0):
    """HBM bytes per launch of `kernel` (FETCH_SIZE * fetch_factor + WRITE_SIZE) read at RUN TIME from the newest
    profiles/r*_bench_infer_pmc_hbm.csv (separate rocprofv3 --pmc passes of this command, tools/collect_profiles.sh;
    KiB per dispatch; fetch_factor 2 = the gfx950 correction for 16-B-per-lane streaming reads).  The most-launched
    instantiation whose name starts with `kernel` is taken.  Returns (bytes, file name); raises SystemExit if the file
    or the kernel is missing -- a renamed kernel must not silently keep a stale number."""
    import csv
    import glob
    import re
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_bench_infer_pmc_hbm.csv")),
                   key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)))
    if not files:
        raise SystemExit("bench.py: no profiles/r*_bench_infer_pmc_hbm.csv to take roofline.traffic from")
    want = kernel.replace(" ", "").rstrip(">")
    best = {}
    for r in csv.DictReader(open(files[-1])):
        name = r["Name"].replace(" ", "")
        if name.startswith(want):
            key = (name, r["Counter"])
            best[key] = (int(r["Calls"]), float(r["AvgValueKiBPerDispatch"]))
    names = {n for n, _ in best}
    if not names:
        raise SystemExit(f"bench.py: kernel {kernel!r} not found in {files[-1]} (re-run tools/collect_profiles.sh)")
    top = max(names, key=lambda n: best.get((n, "FETCH_SIZE"), (0, 0))[0])
    if (top, "FETCH_SIZE") not in best or (top, "WRITE_SIZE") not in best:
        raise SystemExit(f"bench.py: {files[-1]} lacks FETCH_SIZE / WRITE_SIZE rows for {top}")
    kib = best[(top, "FETCH_SIZE")][1] * fetch_factor + best[(top, "WRITE_SIZE")][1]
    return kib * 1024.0, os.path.basename(files[-1])


def encoder_decoder_flops(side: int):
    """Algorithmic FLOPs (2*MAC) per image of (encoder, decoder) (126.54 / 126.53 GF @512)."""
    from oracle.ref_ops import DECODER_CONVS, DECODER_UP_AFTER, ENCODER_CONVS, ENCODER_POOL_AFTER
    enc, h = 0.0, side
    for idx, cin, cout, k in ENCODER_CONVS:
        enc += 2.0 * h * h * cin * cout * k * k
        if idx in ENCODER_POOL_AFTER:
            h = (h + 1) // 2
    dec = 0.0
    for idx, cin, cout in DECODER_CONVS:
        dec += 2.0 * h * h * cin * cout * 9
        if idx in DECODER_UP_AFTER:
            h *= 2
    return enc, dec


def decoder_upsample_layer_flops(side: int) -> float:
    """Algorithmic FLOPs per image of the three decoder convs that read a nearest x2 upsampled input
    (generator.py:92-93, 105-106, 112-113).  In forward-only fp32 calls they run in the sub-pixel form
    (st_conv3x3_up2_fwd): 16 instead of 36 multiply-adds per source pixel and channel pair, same result."""
    from oracle.ref_ops import DECODER_CONVS, DECODER_UP_AFTER, ENCODER_POOL_AFTER
    h = side
    for _ in ENCODER_POOL_AFTER:
        h = (h + 1) // 2
    f, after_up = 0.0, False
    for idx, cin, cout in DECODER_CONVS:
        if after_up:
            f += 2.0 * h * h * cin * cout * 9
        after_up = idx in DECODER_UP_AFTER
        if after_up:
            h *= 2
    return f


def executed_flops_per_image(side: int, compute: str = "f32") -> float:
    """FLOPs the inference kernels actually EXECUTE per stylised image: the algorithmic 2E + D minus the 20/36 of the
    three upsample layers' multiply-adds that the sub-pixel form does not perform (fp32 forward-only calls, unless
    ST_NO_UP2 is set; the bf16 modes run the direct form)."""
    from style_transfer_amd.generator import _PackedWeightCache
    e_f, d_f = encoder_decoder_flops(side)
    total = 2 * e_f + d_f
    if compute == "f32" and _PackedWeightCache.UP2:
        total -= decoder_upsample_layer_flops(side) * 20.0 / 36.0
    return total


def loss_vgg_flops(side: int, upto: int) -> float:
    from oracle.ref_ops import VGG19_FEATURES
    f, h = 0.0, side
    for idx, (kind, cin, cout) in enumerate(VGG19_FEATURES):
        if idx > upto:
            break
        if kind == "conv":
            f += 2.0 * h * h * cin * cout * 9
        elif kind == "pool":
            h //= 2
    return f


def train_flops_per_image(side: int) -> float:
    """Necessary work of one training step per image (405.95 GF @256, 1623.8 GF @512):
    fwd 2E + D + 2 L@21 + 2 L@25; bwd decoder dgrad + wgrad (2D) + loss-VGG dgrad on the decoded passes."""
    enc, dec = encoder_decoder_flops(side)
    l21, l25 = loss_vgg_flops(side, max(CONTENT_IDX)), loss_vgg_flops(side, max(STYLE_IDX))
    return (2 * enc + dec + 2 * l21 + 2 * l25) + 2 * dec + (l21 + l25)


def host_cpu():
    """(model name, physical cores of the machine, CPUs this process may run on) from /proc/cpuinfo."""
    model, phys = "unknown", set()
    try:
        pid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                phys.add((pid, ln.split(":", 1)[1].strip()))
    except OSError:
        pass
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = os.cpu_count() or 1
    return model, len(phys) or None, allowed


def median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def cpu_baseline_infer(pairs: int = 4, runs: int = 5):
    """Time the CPU oracle on a bounded sample: `pairs` 512x512 content/style pairs, alpha=1; median of `runs`
    (median of >= 5 after a warm-up)."""
    from oracle import ref_ops as R
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    model, phys, allowed = host_cpu()
    enc_sd, dec_sd = R.make_encoder_state(0), R.make_decoder_state(1)
    g = torch.Generator().manual_seed(1234)
    c = torch.rand(pairs, 3, SIDE, SIDE, generator=g)
    s = torch.rand(pairs, 3, SIDE, SIDE, generator=g)
    times = []
    with torch.no_grad():
        R.stylize(c[:1], s[:1], enc_sd, dec_sd)        # warm-up
        for _ in range(runs):
            t0 = time.perf_counter()
            R.stylize(c, s, enc_sd, dec_sd)
            times.append(time.perf_counter() - t0)
    med = median(times)
    return {"value": round(pairs / med, 4), "unit": "img/s", "cores": cores, "kind": "port",
            "cpu_model": model, "physical_cores_of_host": phys, "cpus_allowed": allowed,
            "best": round(pairs / min(times), 4),
            "sample": f"{pairs} content/style pairs at {SIDE}x{SIDE} fp32, alpha=1, median of {runs} runs after one "
                      f"warm-up of oracle.ref_ops.stylize (plain torch CPU, {cores} threads)"}


def cpu_baseline_train(batch: int = 2, runs: int = 5):
    """One full training step (forward, losses, backward, Adam) of the CPU oracle at batch 2, 256x256."""
    from oracle import ref_ops as R
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    enc_sd = R.make_encoder_state(0)
    dec_sd = {k: v.clone().requires_grad_(True) for k, v in R.make_decoder_state(1).items()}
    vgg_sd = R.make_loss_vgg_state(2, upto=max(STYLE_IDX))
    ce = torch.tensor([1e-5], requires_grad=True)
    se = torch.tensor([1e-5], requires_grad=True)
    params = list(dec_sd.values()) + [ce, se]
    opt = torch.optim.Adam(params, lr=1e-4)
    g = torch.Generator().manual_seed(1234)
    c = torch.rand(batch, 3, TRAIN_SIDE, TRAIN_SIDE, generator=g)
    s = torch.rand(batch, 3, TRAIN_SIDE, TRAIN_SIDE, generator=g)
    times = []
    for i in range(runs + 1):
        t0 = time.perf_counter()
        losses = R.train_step_losses(c, s, enc_sd, dec_sd, vgg_sd, ce, se, 0.1, CONTENT_IDX, STYLE_IDX,
                                     1.0, 10.0, 1.0, 10.0, False)
        opt.zero_grad()
        losses[4].backward()
        opt.step()
        if i:
            times.append(time.perf_counter() - t0)
    med = median(times)
    model, phys, allowed = host_cpu()
    return {"value": round(batch / med, 4), "unit": "img/s", "cores": cores, "kind": "port",
            "cpu_model": model, "physical_cores_of_host": phys, "cpus_allowed": allowed,
            "best": round(batch / min(times), 4),
            "sample": f"one training step at batch {batch}, {TRAIN_SIDE}x{TRAIN_SIDE} fp32 (mean/std style loss), "
                      f"median of {runs} after 1 warm-up, oracle.ref_ops.train_step_losses + autograd + Adam "
                      f"(plain torch CPU, {cores} threads)"}


class Ctx:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (the HIP hot path has no CPU fallback)")
        # ST_BENCH_REHEARSAL=1: every rank on cuda:0 over gloo -- a code-path rehearsal of N > 1 on a 1-GPU box
        rehearsal = bool(os.environ.get("ST_BENCH_REHEARSAL"))
        dev_index = 0 if rehearsal else self.local_rank
        if dev_index >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {self.rank} needs GPU {dev_index} but this node shows "
                             f"{torch.cuda.device_count()} (ST_BENCH_REHEARSAL=1 runs every rank on cuda:0 over gloo)")
        torch.cuda.set_device(dev_index)
        self.dev = torch.device("cuda", dev_index)
        self.dist = None
        # ST_FORCE_DP=1 under a launcher: initialise the group (RCCL) even for ONE rank, so that a 1-GPU box can
        # drive the data-parallel legs through the real backend (Trainer.dp_active)
        if self.world > 1 or ("WORLD_SIZE" in os.environ and os.environ.get("ST_FORCE_DP")):
            from style_transfer_amd.parallel import init_process_group
            self.dist, self.dev = init_process_group(rehearsal=rehearsal, local_rank=self.local_rank)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds: float) -> float:
        t = torch.tensor([seconds], dtype=torch.float64, device=self.dev)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def timed(ctx: Ctx, fn, steps: int, warmup: int, collect_kernels: bool):
    from style_transfer_amd import ops
    # clock / allocator pre-warm (not a step: a fresh box runs its first ~1 s about 4 % slow), then W warmup steps
    # (the iteration count is agreed over the ranks: a step may contain collectives)
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    n_pre = int(ctx.max_over_ranks(min(50.0, PREWARM_SECONDS / max(1e-4, time.perf_counter() - t0))))
    for i in range(n_pre):
        fn()
        if os.environ.get("ST_BENCH_WATCHDOG"):
            torch.cuda.synchronize()
            print(f"[rank {ctx.rank}] pre-warm {i + 1}/{n_pre} {time.perf_counter() - t0:.2f}s", flush=True)
    torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    ctx.barrier()
    elapsed = ctx.max_over_ranks(time.perf_counter() - t0)
    # Per-launch HIP events for the roofline objects: the SAME K steps are run once more, right after the
    # timed region, with an event pair around every launch on the launch stream.  They are not recorded
    # inside the timed region itself because the event packets open gaps between launches on some boxes
    # (measured: 308 img/s without them, 284-305 img/s with them, kernel durations identical); the step time
    # of the instrumented repeat is reported next to the clean one.
    records, elapsed_ev = [], None
    if collect_kernels and not os.environ.get("ST_NO_KTIMER"):
        ops.KERNEL_TIMER.start()
        fn()                                             # count the launches of one step
        per_step = len(ops.KERNEL_TIMER.stop())
        ctx.barrier()
        ops.KERNEL_TIMER.start(prealloc=per_step * steps)   # events created before the loop
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        ctx.barrier()
        elapsed_ev = time.perf_counter() - t0
        records = ops.KERNEL_TIMER.stop()
    return elapsed, records, out, elapsed_ev


def adain_back_to_back(ctx: Ctx, args, reps: int = 20) -> float:
    """Average duration (us) of the AdaIN launch of one inference step (relu4_1 features of the batch)."""
    from style_transfer_amd import ops
    n, c, hw = args.batch, 512, args.side // 8
    g = torch.Generator(device=ctx.dev).manual_seed(7)
    sets = [(torch.rand(n, c, hw, hw, device=ctx.dev, generator=g), torch.rand(n, c, hw, hw, device=ctx.dev, generator=g))
            for _ in range(2)]
    for fc, fs in sets:
        ops.adain(fc, fs, 1e-5, 1e-5, 1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()                     # torch's current stream = the stream ops.adain launches on
    for _ in range(reps):
        for fc, fs in sets:
            ops.adain(fc, fs, 1e-5, 1e-5, 1.0)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (2 * reps)


def _b2b_us(fn, reps: int) -> float:
    """Average duration (us) of `fn` over `reps` back-to-back calls between ONE HIP event pair on the launch stream."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def roofline_aux(ctx: Ctx):
    """Roofline objects for the loss kernels north_star names next to the convs: RGB-uv histogram and TV (HBM GB/s)
    and the covariance "Gram" kernel (fp32 MFMA TFLOP/s).  Each is timed back to back through the module API at the
    BASELINE configs[2] shape (B = 8, 256x256: a few MB -- launch-latency-bound, reported as such) and at a shape that
    says something about the kernel (B = 16 at 2048x2048 for the image kernels, 806 MB: nothing stays in the 256 MB
    Infinity Cache; [16, 512, 64, 64] for the covariance).  Algorithmic bytes / FLOPs as in SURVEY 8d."""
    import warnings
    from style_transfer_amd import ops
    from style_transfer_amd.losses import HistLoss, TVLoss, VincentStyleLoss
    dev = ctx.dev
    g = torch.Generator(device=dev).manual_seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hist, tv, vin = HistLoss().histogram_block, TVLoss(), VincentStyleLoss()
    out = {}
    with torch.no_grad():
        rows_h, rows_t = {}, {}
        for tag, b, side, reps in (("configs2", 8, 256, 50), ("large", 16, 2048, 10)):
            x = torch.rand(b, 3, side, side, device=dev, generator=g)
            hb = 4.0 * (x.numel() + b * 3 * 64 * 64)
            us = _b2b_us(lambda: hist(x), reps)
            rows_h[tag] = {"shape": [b, 3, side, side], "avg_call_us": round(us, 1), "mbytes": round(hb / 1e6, 2),
                           "gbs": round(hb / us / 1e3, 1)}
            tb = 4.0 * x.numel()
            us = _b2b_us(lambda: tv(x), reps)
            rows_t[tag] = {"shape": [b, 3, side, side], "avg_call_us": round(us, 1), "mbytes": round(tb / 1e6, 2),
                           "gbs": round(tb / us / 1e3, 1)}
            del x
        for name, rows, kern in (("roofline_hist", rows_h, "hist_prep_kernel + hist_accum_kernel + hist_sum_slabs_kernel + "
                                  "hist_normalize_kernel (one RGBuvHistBlock.forward)"),
                                 ("roofline_tv", rows_t, "tv_rows_partial_kernel + finalize_kernel (one TVLoss.forward)")):
            big = rows["large"]
            out[name] = {"bound": "hbm", "achieved": big["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(big["gbs"] / HBM_PEAK_GBS, 4), "traffic": None, "kernel": kern,
                         "shape": big["shape"], "avg_call_us": big["avg_call_us"], "avg_call_mbytes": big["mbytes"],
                         "at_configs2_shape": dict(rows["configs2"], note="a few MB per call: launch-latency-bound"),
                         "method": "calls back to back between one HIP event pair on the launch stream"}
        # covariance: the five style layers of a configs[2] step (B = 8, 256x256 input), then one large problem
        feats = [torch.randn(8, c, hw, hw, device=dev, generator=g) for c, hw in ((64, 256), (128, 128), (256, 64), (512, 32), (512, 32))]
        means = [ops.mean_std(f)[0].reshape(f.shape[0], f.shape[1]) for f in feats]

        def cov_only(f, m):
            from style_transfer_amd import _lib
            lib = _lib.load()
            b, c, h, w = (int(v) for v in f.shape)
            cov = torch.empty((b, c, c), device=dev)
            ws = torch.empty(lib.st_cov_workspace_bytes(b, c, h * w) // 4, device=dev)

            def call():
                _lib.check(lib.st_cov_fwd(_lib.dev_ptr(f), _lib.dev_ptr(m), _lib.dev_ptr(cov), _lib.dev_ptr(ws), b, c, h * w,
                                          _lib.stream_ptr()), "st_cov_fwd")
            return call
        calls = [cov_only(f, m) for f, m in zip(feats, means)]
        us5 = _b2b_us(lambda: [c() for c in calls], 20)
        fl5 = sum(2.0 * f.shape[0] * f.shape[1] ** 2 * f.shape[2] * f.shape[3] for f in feats)
        fbig = torch.randn(16, 512, 64, 64, device=dev, generator=g)
        mbig = ops.mean_std(fbig)[0].reshape(16, 512)
        usb = _b2b_us(cov_only(fbig, mbig), 10)
        flb = 2.0 * 16 * 512 * 512 * 64 * 64
        tfb = flb / usb / 1e6
        out["roofline_cov"] = {"bound": "mfma", "achieved": round(tfb, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(tfb / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                               "kernel": "cov_fwd_kernel<128> + cov_reduce_kernel (st_cov_fwd)", "shape": [16, 512, 64, 64],
                               "avg_call_us": round(usb, 1), "avg_call_gflop": round(flb / 1e9, 2),
                               "at_configs2_shapes": {"layers": "[8,64,256,256] [8,128,128,128] [8,256,64,64] 2x[8,512,32,32]",
                                                      "five_calls_us": round(us5, 1), "gflop": round(fl5 / 1e9, 2),
                                                      "tflops": round(fl5 / us5 / 1e6, 2),
                                                      "note": "4.3 GFLOP per call: ~1 round of blocks, launch-latency-bound"},
                               "method": "calls back to back between one HIP event pair on the launch stream"}
    return out


def run_inference(ctx: Ctx, args, compute: str = "f32"):
    from oracle import ref_ops as R     # seeded synthetic weights only
    from style_transfer_amd.generator import Decoder, Encoder
    from style_transfer_amd.pipeline import stylize
    from style_transfer_amd.transformer import AdaIN

    dev = ctx.dev
    enc = Encoder(weights=R.make_encoder_state(0)).to(dev).eval()
    dec = Decoder().to(dev).eval()
    dec.load_state_dict(R.make_decoder_state(1))
    ada = AdaIN().to(dev)
    g = torch.Generator().manual_seed(1234 + ctx.rank)
    content = torch.rand(args.batch, 3, args.side, args.side, generator=g).to(dev)   # resident in HBM
    style = torch.rand(args.batch, 3, args.side, args.side, generator=g).to(dev)

    from style_transfer_amd import precision
    with precision.compute_dtype(compute):
        elapsed, records, out, elapsed_ev = timed(ctx, lambda: stylize(enc, dec, ada, content, style, alpha=1.0),
                                                  args.steps, args.warmup, collect_kernels=True)
    elapsed_ev = elapsed_ev or elapsed
    assert torch.isfinite(out).all()
    value = ctx.world * args.batch * args.steps / elapsed
    e_f, d_f = encoder_decoder_flops(args.side)
    per_img = 2 * e_f + d_f
    per_img_exec = executed_flops_per_image(args.side, compute)

    if compute in ("bf16", "bf16x3"):
        x3 = compute == "bf16x3"
        dom_name = DOMINANT_BF16X3 if x3 else DOMINANT_BF16
        allconv = [r for r in records if r["kernel"].startswith("conv3x3_bf16")]
        dom = [r for r in allconv if r["kernel"] == dom_name]
        # bf16x3 issues three bf16 MFMAs per algorithmic multiply-add: the matrix pipe does 3x the layer's FLOPs
        mul = 3.0 if x3 else 1.0
        fl, sec = sum(r["flops"] for r in dom), sum(r["ms"] for r in dom) * 1e-3
        ach = mul * fl / sec / 1e12 if sec > 0 else 0.0
        what = ("fp32 operands split into hi+lo bf16 halves, 3 bf16 MFMAs per product, fp32 accumulate (fp32-grade: "
                "~4e-5 max relative error on the pipeline, tests/test_gpu_bf16x3.py)") if x3 else \
               "bf16 MFMA operands / fp32 accumulate"
        return {"metric": "stylized 512x512 img/s (inference)", "value": round(value, 3), "unit": "img/s",
                "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "dtype": compute,
                "config": {"workload": f"inference batch {args.batch} at {args.side}x{args.side} per GPU, {what} / " +
                                       ("fp32 NCHW activations" if x3 else
                                        "fp32 NCHW at the encoder/AdaIN/decoder boundaries, bf16 channel-group (G8) "
                                        "activations between the convs of each network"), "batch_per_gpu": args.batch, "side": args.side},
                "model_tflops": round(value * per_img / 1e12, 2),
                "model_tflops_executed": round(value * per_img_exec / 1e12, 2),
                "roofline": {"bound": "mfma", "achieved": round(ach, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(ach / BF16_MFMA_PEAK_TFLOPS, 4),
                             "traffic": round(pmc_traffic("conv3x3_bf16_kernel<2,4,2,4,false,false,false,false,true,true")[0])
                             if (not x3 and args.batch == BATCH and args.side == SIDE) else None,
                             "kernel": dom_name,
                             "mfma_flops_per_algorithmic_flop": mul,
                             "launches": len(dom), "avg_launch_us": round(sec / max(1, len(dom)) * 1e6, 2),
                             "all_conv_tflops_algorithmic": round(sum(r["flops"] for r in allconv) /
                                                                  max(1e-12, sum(r["ms"] for r in allconv) * 1e-3) / 1e12, 1)}}
    dom = [r for r in records if r["kernel"] == DOMINANT]
    allconv = [r for r in records if r["kernel"].startswith("conv3x3_f32_kernel")]
    flops = sum(r["flops"] for r in dom)
    secs = sum(r["ms"] for r in dom) * 1e-3
    achieved = flops / secs / 1e12 if secs > 0 else 0.0
    default_shape = args.batch == BATCH and args.side == SIDE
    traffic, traffic_file = pmc_traffic(DOMINANT) if default_shape else (None, None)
    roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "traffic": round(traffic) if traffic else None,
            "traffic_note": f"HBM bytes per launch, FETCH_SIZE + WRITE_SIZE read at run time from profiles/{traffic_file} "
                            "(separate rocprofv3 --pmc passes of this command; most-launched instantiation)",
            "kernel": DOMINANT, "launches": len(dom),
            "avg_launch_us": round(secs / max(1, len(dom)) * 1e6, 2),
            "avg_launch_gflop": round(flops / max(1, len(dom)) / 1e9, 3),
            "all_conv_tflops": round(sum(r["flops"] for r in allconv) /
                                     max(1e-12, sum(r["ms"] for r in allconv) * 1e-3) / 1e12, 2),
            "all_conv_time_frac_of_step": round(sum(r["ms"] for r in allconv) * 1e-3 / elapsed, 4),
            "method": "HIP event pair around every launch on the launch stream, over a REPEAT of the K timed "
                      "steps run right after them -- not the timed loop itself (events inside the timed region open "
                      "launch gaps on some boxes), so all_conv_time_frac_of_step relates two different loops and "
                      "may exceed 1",
            "ms_per_step_instrumented": round(elapsed_ev / args.steps * 1e3, 3)}
    roof["method"] += ("; FLOPs per launch are the EXECUTED ones: the sub-pixel launches of the three upsample layers "
                       "(conv3x3_f32_kernel SUB form) count 16 multiply-adds per source pixel and channel pair, not the 36 "
                       "of the direct form they replace")
    ad = [r for r in records if r["kernel"].startswith("adain")]
    ad_bytes = sum(r["bytes"] for r in ad)
    ad_secs = sum(r["ms"] for r in ad) * 1e-3
    ad_gbs = ad_bytes / ad_secs / 1e9 if ad_secs > 0 else 0.0
    # The in-step figure brackets ONE ~60 us launch with its own event pair (~8 us of event overhead on top of
    # the kernel); the same launch is therefore also timed back to back, 2 x 20 launches between one event
    # pair, alternating two feature sets (2 x 402 MB working set >> the 256 MB Infinity Cache: no cache help).
    b2b_us = adain_back_to_back(ctx, args) if default_shape else None
    if b2b_us:
        ad_gbs_step, ad_gbs = ad_gbs, ad_bytes / max(1, len(ad)) / (b2b_us * 1e-6) / 1e9
    else:
        ad_gbs_step = ad_gbs
    roof_adain = {"bound": "hbm", "achieved": round(ad_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": round(ad_gbs / HBM_PEAK_GBS, 4), "kernel": "adain_reg_kernel<4>", "launches": len(ad),
                  "avg_launch_us": round(b2b_us, 2) if b2b_us else round(ad_secs / max(1, len(ad)) * 1e6, 2),
                  "method": "40 launches back to back between one HIP event pair on the launch stream, two "
                            "alternating feature sets" if b2b_us else "per-launch HIP event pairs inside the timed steps",
                  "in_step_avg_launch_us": round(ad_secs / max(1, len(ad)) * 1e6, 2),
                  "in_step_achieved": round(ad_gbs_step, 1),
                  "avg_launch_mbytes": round(ad_bytes / max(1, len(ad)) / 1e6, 2),
                  "traffic": round(ad_bytes / max(1, len(ad))) if default_shape else None,
                  "traffic_note": "algorithmic bytes; the PMC passes agree: 2 x FETCH_SIZE (gfx950 16-B/lane "
                                  "correction) + WRITE_SIZE of adain_reg_kernel in profiles/*_bench_infer_pmc_hbm.csv"}
    return {
        "metric": "stylized 512x512 img/s (inference)", "value": round(value, 3), "unit": "img/s",
        "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: inference batch {args.batch} at {args.side}x{args.side} "
                               "fp32 per GPU, 2x VGG19-relu4_1 encoder + AdaIN + decoder, alpha=1",
                   "batch_per_gpu": args.batch, "side": args.side,
                   "parallelism": f"images sharded x{ctx.world}, no collective",
                   "gflop_per_image": round(per_img / 1e9, 2),
                   "executed_gflop_per_image": round(per_img_exec / 1e9, 2)},
        "model_tflops": round(value * per_img / 1e12, 2),
        "model_tflops_executed": round(value * per_img_exec / 1e12, 2),
        "model_tflops_note": "model_tflops counts the algorithmic 2E + D FLOPs per image (SURVEY 8d); "
                             "model_tflops_executed counts what the kernels perform (sub-pixel upsample layers: 16/36)",
        "roofline": roof, "roofline_adain": roof_adain,
    }


def run_hires(ctx: Ctx, steps: int = 3, warmup: int = 1, side: int = 2048):
    """BASELINE configs[4]: one 2048x2048 pair per GPU, alpha in {0.5, 1, 1.5} (three stylisations per step), fp32.
    256 KB feature planes: the AdaIN launches are adain_big_kernel (one plane per block held in registers)."""
    from oracle import ref_ops as R     # seeded synthetic weights only
    from style_transfer_amd import ops
    from style_transfer_amd.generator import Decoder, Encoder
    from style_transfer_amd.pipeline import stylize
    from style_transfer_amd.transformer import AdaIN

    dev = ctx.dev
    enc = Encoder(weights=R.make_encoder_state(0)).to(dev).eval()
    dec = Decoder().to(dev).eval()
    dec.load_state_dict(R.make_decoder_state(1))
    ada = AdaIN().to(dev)
    g = torch.Generator().manual_seed(4321 + ctx.rank)
    content = torch.rand(1, 3, side, side, generator=g).to(dev)
    style = torch.rand(1, 3, side, side, generator=g).to(dev)
    alphas = (0.5, 1.0, 1.5)

    def step():
        out = None
        for a in alphas:
            out = stylize(enc, dec, ada, content, style, alpha=a)
        return out
    for _ in range(warmup):
        step()
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    ctx.barrier()
    elapsed = ctx.max_over_ranks(time.perf_counter() - t0)
    assert torch.isfinite(out).all()
    ops.KERNEL_TIMER.start()
    step()
    recs = [r for r in ops.KERNEL_TIMER.stop() if r["kernel"] == "adain_fwd"]
    n_img = ctx.world * len(alphas) * steps
    e_f, d_f = encoder_decoder_flops(side)
    per_img = 2 * e_f + d_f
    per_img_exec = executed_flops_per_image(side, "f32")
    ad_us = sum(r["ms"] for r in recs) / max(1, len(recs)) * 1e3
    ad_bytes = recs[0]["bytes"] if recs else 0.0
    return {"metric": "stylized 2048x2048 img/s (inference)", "value": round(n_img / elapsed, 3), "unit": "img/s",
            "n_gpus": ctx.world, "steps": steps, "warmup": warmup, "ms_per_image": round(elapsed / (len(alphas) * steps) * 1e3, 2),
            "dtype": "f32", "scaling": "weak",
            "config": {"workload": f"BASELINE configs[4]: one {side}x{side} content/style pair per GPU, alpha sweep "
                                   "0.5 / 1 / 1.5 (three stylisations per step), fp32, images sharded with no collective",
                       "batch_per_gpu": 1, "side": side, "gflop_per_image": round(per_img / 1e9, 1),
                       "executed_gflop_per_image": round(per_img_exec / 1e9, 1)},
            "model_tflops": round(n_img / elapsed * per_img / 1e12, 2),
            "model_tflops_executed": round(n_img / elapsed * per_img_exec / 1e12, 2),
            "roofline_adain": {"bound": "hbm", "kernel": "adain_big_kernel (256 KB planes)",
                               "achieved": round(ad_bytes / max(1e-9, ad_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ad_bytes / max(1e-9, ad_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                               "avg_launch_us": round(ad_us, 2), "avg_launch_mbytes": round(ad_bytes / 1e6, 2),
                               "method": "HIP event pair around each of the step's three launches (an event pair costs "
                                         "about 8 us on a launch this short)"}}


def run_train(ctx: Ctx, args, use_vincent: bool, compute: str = "f32", batch: int = None, side: int = None,
              hip_graph: bool = False):
    from oracle import ref_ops as R     # seeded synthetic weights only
    from style_transfer_amd.trainer import GradBucket, GraphedStep, Trainer

    dev = ctx.dev
    import contextlib
    with warnings.catch_warnings(), contextlib.redirect_stdout(sys.stderr):   # stdout carries ONLY the JSON line
        warnings.simplefilter("ignore")
        tr = Trainer({"train": []}, output_dir="/tmp/st_bench", num_train_epochs=1,
                     per_device_batch_size=args.train_batch, learning_rate=1e-4, alpha=1.0, beta=10.0, gamma=1.0,
                     delta=10.0, eps=0.1, do_decoder_train=True, optim_name={"optim_name": "adam"},
                     use_VincentStyleLoss=use_vincent, content_layers_idx=CONTENT_IDX, style_layers_idx=STYLE_IDX,
                     encoder_weights=R.make_encoder_state(0), vgg_weights=R.make_loss_vgg_state(2, upto=max(STYLE_IDX)),
                     device=dev, use_hip_graph=hip_graph)
        encoder, decoder, transformer, loss_vgg = tr.build_model()
        decoder.load_state_dict(R.make_decoder_state(1))
        params = tr.trainable_parameters(transformer, decoder)
        hip_graph = hip_graph and tr.graph_capable()
        lr = torch.tensor(1e-4, device=dev) if hip_graph else 1e-4
        opt = tr.get_optimizer("adam", params, lr, tr.graph_optim_kwargs())
        step = tr.make_step(encoder, decoder, transformer, loss_vgg)
    bucket = GradBucket(params)
    graphed = GraphedStep(tr, step, opt, bucket) if hip_graph else None
    g = torch.Generator().manual_seed(4321 + ctx.rank)
    s = side or args.train_side
    tb = batch or args.train_batch
    content = torch.rand(tb, 3, s, s, generator=g).to(dev)
    style = torch.rand(tb, 3, s, s, generator=g).to(dev)

    def one():
        if graphed is not None:
            return graphed(content, style)      # 3 eager warm-up calls, capture, then replays (all before the timed loop)
        # Gram variant: the trainer's one-batch lookahead (Trainer.train) -- the NEXT batch's weight-independent
        # forward is issued under this batch's eigen-decompositions; here the next batch is the same synthetic pair,
        # its front is recomputed every step like any other batch's
        return tr.optimize_step(step, opt, bucket, content, style, prefetch=(content, style) if use_vincent else None)

    from style_transfer_amd import precision
    steps, warmup = args.train_steps, args.train_warmup
    with precision.compute_dtype(compute):
        if graphed is not None:
            for _ in range(GraphedStep.WARMUP + 1):     # eager warm-up calls + the capture: never inside the timed loop
                one()
            assert graphed.graph is not None
        elapsed, _, losses, _ = timed(ctx, one, steps, warmup, collect_kernels=False)
    total = float(losses[4].detach())
    assert total == total, "training loss is NaN"
    value = ctx.world * tb * steps / elapsed
    per_img = train_flops_per_image(s)
    return {"metric": "train img/s", "value": round(value, 3), "unit": "img/s", "n_gpus": ctx.world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "scaling": "weak",
            "dtype": compute, "style_loss": "VincentStyleLoss (covariance MFMA + eigh)" if use_vincent
            else "AdaINStyleLoss (mean/std)",
            "launch": (("eager launches" + (", next batch's weight-independent forward issued under the "
                                            "eigen-decompositions (one-batch lookahead)" if use_vincent else ""))
                       if graphed is None else "one HIP graph replay per step (GraphedStep)"
                       if not tr.dp_active else "three HIP graph replays per step around the two eager all-reduces "
                                                "(GraphedStep, data parallel)"),
            "config": {"workload": ("BASELINE configs[2]: " if (compute, tb, s) == ("f32", 8, 256) else "") +
                                   f"training step batch {tb} at {s}x{s} {compute} per "
                                   "GPU: 2x encoder, AdaIN, decoder, Normalize, loss VGG19 (idx 21 / 2,7,14,19,25), "
                                   "content + style + TV + RGB-uv histogram losses, backward, Adam(decoder + eps)",
                       "batch_per_gpu": tb, "side": s,
                       "parallelism": f"dp{ctx.world}: batch sharded, one 14.02 MB fp32 RCCL all-reduce + one 2-float "
                                      "all-reduce per step" if tr.dp_active else "single GPU",
                       "gflop_per_image_necessary": round(per_img / 1e9, 2)},
            "model_tflops": round(value * per_img / 1e12, 2), "final_total_loss": round(total, 4)}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves, exactly as the driver
    would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), as a CHILD
    process -- this parent never initialises the GPU, it only forwards rank 0's JSON line and the exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=REPO)
    for ln in proc.stdout:                      # stdout carries ONE JSON line (rank 0); anything else goes to stderr
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--side", type=int, default=SIDE)
    ap.add_argument("--mode", choices=["both", "infer", "train"], default="both")
    ap.add_argument("--train-batch", type=int, default=TRAIN_BATCH)
    ap.add_argument("--train-side", type=int, default=TRAIN_SIDE)
    ap.add_argument("--train-steps", type=int, default=10)
    ap.add_argument("--train-warmup", type=int, default=3)
    ap.add_argument("--no-gram", action="store_true", help="skip the VincentStyleLoss variant of the train leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the roofline_hist / roofline_tv / roofline_cov objects")
    ap.add_argument("--no-hip-graph", action="store_true", help="skip the HIP-graph-replay variants of the train legs")
    ap.add_argument("--compute", choices=["f32", "bf16", "bf16x3"], default="f32",
                    help="conv compute precision of the main legs (profiling aid; the headline legs are f32)")
    ap.add_argument("--no-hires", action="store_true", help="skip the 2048x2048 alpha-sweep leg (BASELINE configs[4])")
    ap.add_argument("--no-bf16", action="store_true",
                    help="skip the bf16 (BASELINE configs[3] shape) and bf16x3 compute-mode legs")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)                # nothing in this process has touched the GPU yet
    if os.environ.get("ST_BENCH_WATCHDOG"):      # dump every thread's stack and exit if the run hangs
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ST_BENCH_WATCHDOG"]), exit=True)

    ctx = Ctx()
    line = {}

    def extra(fn, *a, **kw):
        # Secondary legs under a launcher (N > 1): a failure is recorded in the line instead of taking the headline
        # (the inference and fp32 training legs above them) down with it.  N = 1: errors surface as usual.
        if ctx.world == 1:
            return fn(*a, **kw)
        try:
            return fn(*a, **kw)
        except Exception as e:      # noqa: BLE001
            torch.cuda.synchronize()
            return {"error": f"{type(e).__name__}: {e}"[:300]}

    if args.mode in ("both", "infer"):
        line = run_inference(ctx, args, compute=args.compute)
    if args.mode in ("both", "train"):
        train = run_train(ctx, args, use_vincent=False, compute=args.compute)
        if ctx.world == 1 and not args.no_hip_graph:
            # the same step replayed from a HIP graph (N > 1: the graphed data-parallel leg runs last, see below)
            g = run_train(ctx, args, use_vincent=False, compute=args.compute, hip_graph=True)
            train["hip_graph"] = {k: g[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "launch")}
        if not args.no_gram:
            train["gram"] = extra(run_train, ctx, args, use_vincent=True, compute=args.compute)
        if args.mode == "train":
            line = dict(train, higher_is_better=True, vs_baseline=None, data="synthetic")
            train = None
        if train is not None:
            line["train"] = train
    if not args.no_bf16:
        # BASELINE configs[3] per-GPU shape (batch 64 over 8 GPUs = 8 per GPU at 512x512, bf16 compute)
        bf = {}
        if args.mode in ("both", "infer"):
            bf["inference"] = extra(run_inference, ctx, args, compute="bf16")
        if args.mode in ("both", "train"):
            bf["train"] = extra(run_train, ctx, args, use_vincent=False, compute="bf16", batch=8, side=512)
            if "config" in bf["train"]:
                bf["train"]["config"]["workload"] = ("BASELINE configs[3] per-GPU shape: " +
                                                     bf["train"]["config"]["workload"])
        line["bf16"] = bf
        # fp32-grade results from the bf16 matrix pipe (hi+lo operand split, 3 MFMAs per product); same shapes as
        # the headline legs (configs[1] inference, configs[2] training; weight gradient on the exact fp32 kernel)
        x3 = {}
        if args.mode in ("both", "infer"):
            x3["inference"] = extra(run_inference, ctx, args, compute="bf16x3")
        if args.mode in ("both", "train"):
            x3["train"] = extra(run_train, ctx, args, use_vincent=False, compute="bf16x3",
                                hip_graph=(ctx.world == 1 and not args.no_hip_graph))
        line["bf16x3"] = x3
    if ctx.world > 1 and args.mode in ("both", "train") and not args.no_hip_graph:
        # Data-parallel step replayed from three HIP graphs around the two eager all-reduces.  Last leg of the run and
        # guarded: the eager legs above are the headline, this one must not be able to take the line down with it.
        tgt = line["train"] if args.mode == "both" else line
        try:
            g = run_train(ctx, args, use_vincent=False, compute=args.compute, hip_graph=True)
            tgt["hip_graph"] = {k: g[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "launch")}
        except Exception as e:      # noqa: BLE001
            tgt["hip_graph"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if args.mode in ("both", "infer") and not args.no_hires and line:
        line["hires"] = extra(run_hires, ctx)
    if ctx.world == 1 and args.mode == "both" and not args.no_aux:
        line.update(roofline_aux(ctx))
    if ctx.rank == 0:
        if ctx.world == 1 and not args.no_cpu_baseline:
            if args.mode in ("both", "infer"):
                line["cpu_baseline"] = cpu_baseline_infer()
            if args.mode in ("both", "train"):
                tgt = line["train"] if args.mode == "both" else line
                tgt["cpu_baseline"] = cpu_baseline_train()
        print(json.dumps(line), flush=True)
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
