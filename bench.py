#!/usr/bin/env python3
"""Headline benchmark: stylised 512x512 images per second (inference), BASELINE.json configs[1]
(batch 16, fp32, one MI355X); for N > 1 every rank stylises its own batch of 16 (images are
independent: no data-path collective, weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task brief) with two extra objects:
  roofline     -- the dominant kernel (fp32 MFMA conv3x3, 128-channel tile instantiation):
                  algorithmic FLOPs per launch / average launch duration, measured with HIP
                  events on the launch stream inside the timed region;
  cpu_baseline -- the CPU oracle (oracle/ref_ops.py, a port of the reference's torch path)
                  timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 FLOP/clk/CU x 256 CU x 2.4 GHz
BATCH = 16
SIDE = 512


def encoder_decoder_flops(side: int) -> float:
    """Algorithmic FLOPs (2*MAC) per stylised image: 2 x encoder + decoder (379.61 GF @512)."""
    from oracle.ref_ops import DECODER_CONVS, DECODER_UP_AFTER, ENCODER_CONVS, ENCODER_POOL_AFTER
    enc = 0.0
    h = side
    for idx, cin, cout, k in ENCODER_CONVS:
        enc += 2.0 * h * h * cin * cout * k * k
        if idx in ENCODER_POOL_AFTER:
            h = (h + 1) // 2
    dec = 0.0
    for idx, cin, cout in DECODER_CONVS:
        dec += 2.0 * h * h * cin * cout * 9
        if idx in DECODER_UP_AFTER:
            h *= 2
    return 2 * enc + dec


def cpu_baseline(pairs: int = 2, runs: int = 2):
    """Time the CPU oracle on a bounded sample: `pairs` 512x512 content/style pairs, alpha=1."""
    from oracle import ref_ops as R
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
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
    best = min(times)
    return {"value": round(pairs / best, 4), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": f"{pairs} content/style pairs at {SIDE}x{SIDE} fp32, alpha=1, best of {runs} runs "
                      f"of oracle.ref_ops.stylize (plain torch CPU, {cores} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--side", type=int, default=SIDE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from oracle import ref_ops as R     # seeded synthetic weights only
    from style_transfer_amd import ops
    from style_transfer_amd.generator import Decoder, Encoder
    from style_transfer_amd.pipeline import stylize
    from style_transfer_amd.transformer import AdaIN

    enc = Encoder(weights=R.make_encoder_state(0)).to(dev).eval()
    dec = Decoder().to(dev).eval()
    dec.load_state_dict(R.make_decoder_state(1))
    ada = AdaIN().to(dev)

    g = torch.Generator().manual_seed(1234 + rank)
    content = torch.rand(args.batch, 3, args.side, args.side, generator=g).to(dev)   # resident in HBM
    style = torch.rand(args.batch, 3, args.side, args.side, generator=g).to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        stylize(enc, dec, ada, content, style, alpha=1.0)
    barrier()
    ops.KERNEL_TIMER.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = stylize(enc, dec, ada, content, style, alpha=1.0)
    barrier()
    elapsed = time.perf_counter() - t0
    records = ops.KERNEL_TIMER.stop()
    assert torch.isfinite(out).all()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    total_imgs = world * args.batch * args.steps
    value = total_imgs / elapsed

    if rank == 0:
        # dominant kernel: the 128-channel-tile instantiation of the fp32 MFMA conv
        dom = [r for r in records if r["kernel"] == "conv3x3_f32_kernel<2,2,2,2>"]
        allconv = [r for r in records if r["kernel"].startswith("conv3x3_f32_kernel")]
        flops = sum(r["flops"] for r in dom)
        secs = sum(r["ms"] for r in dom) * 1e-3
        achieved = flops / secs / 1e12 if secs > 0 else 0.0
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "kernel": "conv3x3_f32_kernel<2,2,2,2>", "launches": len(dom),
                "avg_launch_us": round(secs / max(1, len(dom)) * 1e6, 2),
                "avg_launch_gflop": round(flops / max(1, len(dom)) / 1e9, 3),
                "all_conv_tflops": round(sum(r["flops"] for r in allconv) /
                                         max(1e-12, sum(r["ms"] for r in allconv) * 1e-3) / 1e12, 2),
                "all_conv_time_frac_of_step": round(sum(r["ms"] for r in allconv) * 1e-3 / elapsed, 4)}
        line = {
            "metric": "stylized 512x512 img/s (inference)", "value": round(value, 3), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: inference batch {args.batch} at {args.side}x{args.side} "
                                   "fp32 per GPU, 2x VGG19-relu4_1 encoder + AdaIN + decoder, alpha=1",
                       "batch_per_gpu": args.batch, "side": args.side, "parallelism": f"images sharded x{world}, "
                       "no collective", "gflop_per_image": round(encoder_decoder_flops(args.side) / 1e9, 2)},
            "model_tflops": round(value * encoder_decoder_flops(args.side) / 1e12, 2),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
