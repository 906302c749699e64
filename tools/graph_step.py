#!/usr/bin/env python3
"""Experiment: capture one training step in a HIP graph (torch.cuda.CUDAGraph) and replay it.
    python tools/graph_step.py [f32|bf16|bf16x3] [side] [batch]"""
import contextlib
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from style_transfer_amd import precision  # noqa: E402


def main():
    compute = sys.argv[1] if len(sys.argv) > 1 else "f32"
    side = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    from oracle import ref_ops as R
    from style_transfer_amd.trainer import GradBucket, Trainer
    dev = torch.device("cuda", 0)
    with warnings.catch_warnings(), contextlib.redirect_stdout(sys.stderr):
        warnings.simplefilter("ignore")
        tr = Trainer({"train": []}, output_dir="/tmp/st_bench", num_train_epochs=1, per_device_batch_size=batch,
                     learning_rate=1e-4, alpha=1.0, beta=10.0, gamma=1.0, delta=10.0, eps=0.1, do_decoder_train=True,
                     optim_name={"optim_name": "adam", "capturable": True}, use_VincentStyleLoss=False,
                     content_layers_idx=bench.CONTENT_IDX, style_layers_idx=bench.STYLE_IDX,
                     encoder_weights=R.make_encoder_state(0),
                     vgg_weights=R.make_loss_vgg_state(2, upto=max(bench.STYLE_IDX)), device=dev)
        encoder, decoder, transformer, loss_vgg = tr.build_model()
        params = tr.trainable_parameters(transformer, decoder)
        opt = tr.get_optimizer("adam", params, 1e-4, {"optim_name": "adam", "capturable": True})
        step = tr.make_step(encoder, decoder, transformer, loss_vgg)
    bucket = GradBucket(params)
    c = torch.rand(batch, 3, side, side, device=dev)
    s = torch.rand(batch, 3, side, side, device=dev)
    with precision.compute_dtype(compute):
        side_stream = torch.cuda.Stream()
        side_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side_stream):
            for _ in range(3):
                eager = tr.optimize_step(step, opt, bucket, c, s)
        torch.cuda.current_stream().wait_stream(side_stream)
        torch.cuda.synchronize()
        print("eager total loss", float(eager[4]), flush=True)
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            losses = tr.optimize_step(step, opt, bucket, c, s)
        torch.cuda.synchronize()
        print("captured", flush=True)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print("replayed total loss", float(losses[4]), flush=True)
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{compute} B={batch} {side}^2 graph replay: host {1e3 * (t1 - t0) / n:.2f} ms/step, "
          f"total {1e3 * (t2 - t0) / n:.2f} ms/step", flush=True)


if __name__ == "__main__":
    main()
