"""Data-parallel training step on the REAL engine (Trainer.optimize_step / GraphedStep on the HIP kernels) with
2 ranks: both ranks on cuda:0, collectives over gloo staged through the host (style_transfer_amd.parallel,
rehearsal mode -- a 1-GPU box cannot run RCCL with 2 ranks).  SURVEY.md 8e: batch sharded by rank, ONE flat gradient
bucket all-reduced per step, the two Hellinger sums all-reduced in the forward (HistLoss's sqrt sits outside the
batch sum), AdaIN gradient clamp after the reduction.  The reduced gradients, the losses and the weights after
several optimizer steps must equal a single-process step of the same HIP path on the whole batch."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIDE, PER_RANK, WORLD, STEPS = 64, 2, 2, 6
TINY_LR = 0.0          # Adam still runs (state, step counters), the weights stay bit-identical


def _make(world_size_override=None, use_graph=False, threshold=2e-3, lr_value=1e-4):
    import warnings
    from oracle import ref_ops as R
    from style_transfer_amd.trainer import GradBucket, GraphedStep, Trainer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr = Trainer({"train": []}, output_dir="/tmp/st_dp_test", num_train_epochs=1, per_device_batch_size=PER_RANK,
                     learning_rate=lr_value, alpha=1.0, beta=10.0, gamma=1.0, delta=10.0, eps=0.1, do_decoder_train=True,
                     optim_name={"optim_name": "adam"}, content_layers_idx=[21], style_layers_idx=[2, 7, 14, 19, 25],
                     encoder_weights=R.make_encoder_state(0), vgg_weights=R.make_loss_vgg_state(2, upto=25),
                     gradient_threshold=threshold, device="cuda", use_hip_graph=True)
        if world_size_override is not None:
            tr.world_size = world_size_override
        encoder, decoder, transformer, loss_vgg = tr.build_model()
        decoder.load_state_dict(R.make_decoder_state(1))
        params = tr.trainable_parameters(transformer, decoder)
        lr = torch.tensor(lr_value, device="cuda")
        opt = tr.get_optimizer("adam", params, lr, tr.graph_optim_kwargs())
        step = tr.make_step(encoder, decoder, transformer, loss_vgg)
    bucket = GradBucket(params)
    run = GraphedStep(tr, step, opt, bucket) if use_graph else \
        (lambda c, s: tr.optimize_step(step, opt, bucket, c, s))
    return tr, run, bucket, decoder, transformer


def _batches():
    g = torch.Generator().manual_seed(21)
    return [(torch.rand(PER_RANK * WORLD, 3, SIDE, SIDE, generator=g), torch.rand(PER_RANK * WORLD, 3, SIDE, SIDE, generator=g))
            for _ in range(STEPS)]


def _trajectory(run, bucket, decoder, transformer, batches, lo, hi):
    grads, losses = [], []
    for c, s in batches:
        out = run(c[lo:hi].cuda().contiguous(), s[lo:hi].cuda().contiguous())
        grads.append(bucket.flat.detach().clone())
        losses.append(torch.stack([l.detach().reshape(()) for l in out]).cpu())
    torch.cuda.synchronize()
    weights = torch.cat([p.detach().flatten() for p in list(decoder.parameters()) + list(transformer.parameters())])
    return [g.cpu() for g in grads], losses, weights.cpu()


RUNS = (("frozen", False, TINY_LR), ("eager", False, 1e-4), ("graph", True, 1e-4))


def _worker(rank, world, port, q, compute="f32", runs=RUNS, per_rank=PER_RANK):
    sys.path.insert(0, REPO)
    from style_transfer_amd import precision
    from style_transfer_amd.parallel import init_process_group
    precision.set_compute_dtype(compute)
    dist, _ = init_process_group(rehearsal=True, local_rank=rank, init_method=f"tcp://127.0.0.1:{port}",
                                 rank=rank, world_size=world)
    batches = _batches()
    lo = rank * per_rank
    result = {}
    # "frozen": lr = 0, the weights stay put, so EVERY step compares gradients of identical
    # weights on a new batch; the other two runs take real Adam steps (lr 1e-4), eager and graph-replayed
    for name, use_graph, lr_value in runs:
        tr, run, bucket, decoder, transformer = _make(use_graph=use_graph, lr_value=lr_value)
        assert tr.world_size == world and tr.graph_capable()
        grads, losses, weights = _trajectory(run, bucket, decoder, transformer, batches, lo, lo + per_rank)
        if use_graph:
            assert len(run.graphs) == 3 and run.replays == STEPS - run.WARMUP
        # every rank's losses (content / style / variation are per-rank means, hist is already batch-wide)
        all_losses = [torch.zeros(STEPS, 5) for _ in range(world)]
        dist.all_gather(all_losses, torch.stack(losses))
        result[name] = {"grads": [g.numpy() for g in grads],
                                                     "losses": [l.numpy() for l in all_losses],
                                                     "weights": weights.numpy()}
    if rank == 0:
        q.put(result)
    dist.barrier()
    dist.destroy_process_group()


def _rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _two_ranks(port, compute="f32", runs=RUNS, world=WORLD, per_rank=PER_RANK):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, compute, runs, per_rank)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        both = q.get(timeout=400)
    finally:
        for p in procs:
            p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs)
    return both


def test_two_rank_step_equals_large_batch_step():
    both = _two_ranks(29600 + os.getpid() % 2000)
    got = both["frozen"]

    # (1) 2 ranks x 2 images == single process, whole batch, same HIP path (world_size 1: no collectives).  Weights
    # are the same in both runs at every step (TINY_LR), so only the summation order of a 4-image launch vs two
    # 2-image launches differs: the exactness statement of SURVEY 8e, on six different batches.
    tr, run, bucket, decoder, transformer = _make(world_size_override=1, use_graph=False, lr_value=TINY_LR)
    grads, losses, weights = _trajectory(run, bucket, decoder, transformer, _batches(), 0, PER_RANK * WORLD)
    for step in range(STEPS):
        ltol, gtol = 2e-5, 2e-4
        ref_l = losses[step].double()
        dp = torch.stack([torch.as_tensor(l[step]).double() for l in got["losses"]])
        # per-sample-mean terms: the mean over ranks is the large-batch value; hist: every rank holds the global one
        for i, name in enumerate(("content", "style", "variation")):
            assert float(dp[:, i].mean()) == pytest.approx(float(ref_l[i]), rel=ltol), (step, name)
        for r in range(WORLD):
            assert float(dp[r, 3]) == pytest.approx(float(ref_l[3]), rel=ltol), (step, "hist", r)
        assert float(dp[:, 4].mean()) == pytest.approx(float(ref_l[4]), rel=ltol), (step, "total")
        # the reduced (mean) gradient bucket == the large-batch gradient, incl. the 2 clamped AdaIN eps gradients
        assert _rel_l2(got["grads"][step], grads[step]) < gtol, step
        g_eps, r_eps = torch.as_tensor(got["grads"][step][:2]), grads[step][:2]
        assert torch.allclose(g_eps, r_eps, rtol=10 * gtol, atol=1e-7), (step, g_eps, r_eps)
        assert float(g_eps.abs().max()) <= 2e-3 + 1e-9                     # clamp applied (after the reduction)
    assert _rel_l2(got["weights"], weights) < 1e-6

    # (2) the same 2-rank run replayed from three HIP graphs around the two eager all-reduces walks the SAME
    # trajectory as the eager 2-rank run (same launches, same partitioning: rounding-identical)
    gr, got = both["graph"], both["eager"]
    for step in range(STEPS):
        for r in range(WORLD):
            a, b = torch.as_tensor(gr["losses"][r][step]).double(), torch.as_tensor(got["losses"][r][step]).double()
            assert torch.allclose(a, b, rtol=1e-5, atol=0), (step, r, a, b)
        assert _rel_l2(gr["grads"][step], got["grads"][step]) < 1e-5, step
    assert _rel_l2(gr["weights"], got["weights"]) < 1e-5


def test_four_rank_step_equals_large_batch_step():
    """4 ranks x 1 image (the same four-image batches): the reduced gradient bucket, the clamped eps gradients and the
    losses equal the single-process 4-image step, eagerly and replayed from the three HIP graphs around the two
    collectives -- more ranks than the 2-rank test, the smallest per-rank batch."""
    world, per_rank = 4, 1
    both = _two_ranks(29800 + os.getpid() % 2000, "f32", (("frozen", False, TINY_LR), ("eager", False, 1e-4), ("graph", True, 1e-4)),
                      world=world, per_rank=per_rank)
    got = both["frozen"]
    tr, run, bucket, decoder, transformer = _make(world_size_override=1, use_graph=False, lr_value=TINY_LR)
    grads, losses, weights = _trajectory(run, bucket, decoder, transformer, _batches(), 0, per_rank * world)
    for step in range(STEPS):
        ref_l = losses[step].double()
        dp = torch.stack([torch.as_tensor(l[step]).double() for l in got["losses"]])
        assert dp.shape[0] == world
        for i, name in enumerate(("content", "style", "variation")):
            assert float(dp[:, i].mean()) == pytest.approx(float(ref_l[i]), rel=2e-5), (step, name)
        for r in range(world):
            assert float(dp[r, 3]) == pytest.approx(float(ref_l[3]), rel=2e-5), (step, "hist", r)
        assert _rel_l2(got["grads"][step], grads[step]) < 2e-4, step
        assert float(torch.as_tensor(got["grads"][step][:2]).abs().max()) <= 2e-3 + 1e-9
    assert _rel_l2(got["weights"], weights) < 1e-6
    gr, eg = both["graph"], both["eager"]
    for step in range(STEPS):
        assert _rel_l2(gr["grads"][step], eg["grads"][step]) < 1e-5, step
    assert _rel_l2(gr["weights"], eg["weights"]) < 1e-5


def test_two_rank_step_equals_large_batch_step_in_bf16_compute_mode():
    """BASELINE configs[3] is data parallel in the bf16 compute mode: the same exactness statement there.  Every
    image's convolutions round identically whatever the batch they run in (bf16 operands, fp32 accumulation per
    output element), so the 2-rank step differs from the 4-image step only in fp32 summation order, as in fp32."""
    from style_transfer_amd import precision
    got = _two_ranks(29650 + os.getpid() % 2000, "bf16", (("frozen", False, TINY_LR),))["frozen"]
    precision.set_compute_dtype("bf16")
    try:
        tr, run, bucket, decoder, transformer = _make(world_size_override=1, use_graph=False, lr_value=TINY_LR)
        grads, losses, weights = _trajectory(run, bucket, decoder, transformer, _batches(), 0, PER_RANK * WORLD)
    finally:
        precision.set_compute_dtype("f32")
    for step in range(STEPS):
        ref_l = losses[step].double()
        dp = torch.stack([torch.as_tensor(l[step]).double() for l in got["losses"]])
        for i, name in enumerate(("content", "style", "variation")):
            assert float(dp[:, i].mean()) == pytest.approx(float(ref_l[i]), rel=1e-4), (step, name)
        for r in range(WORLD):
            assert float(dp[r, 3]) == pytest.approx(float(ref_l[3]), rel=1e-4), (step, "hist", r)
        assert _rel_l2(got["grads"][step], grads[step]) < 1e-3, (step, _rel_l2(got["grads"][step], grads[step]))
    assert _rel_l2(got["weights"], weights) < 1e-6
    # ... and it WAS the bf16 mode: the fp32 step's gradient is a visibly different number
    tr, run, bucket, decoder, transformer = _make(world_size_override=1, use_graph=False, lr_value=TINY_LR)
    g32, _, _ = _trajectory(run, bucket, decoder, transformer, _batches()[:1], 0, PER_RANK * WORLD)
    assert _rel_l2(grads[0], g32[0]) > 1e-3


# ---- the real RCCL backend, one rank ---------------------------------------------------------------------------
def _rccl_worker(port, q):
    """ONE rank on backend "nccl" (= RCCL) with ST_FORCE_DP=1: the data-parallel code path -- both all-reduces per
    step, the three-graph replay with eager collectives between the replays, graph capture next to the process
    group's watchdog thread -- runs through the real backend on the real device."""
    sys.path.insert(0, REPO)
    os.environ["ST_FORCE_DP"] = "1"
    from style_transfer_amd.parallel import init_process_group
    dist, _ = init_process_group(rehearsal=False, local_rank=0, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    batches = _batches()
    result = {}
    for name, use_graph in (("eager", False), ("graph", True)):
        tr, run, bucket, decoder, transformer = _make(use_graph=use_graph)
        assert tr.world_size == 1 and tr.dp_active
        grads, losses, weights = _trajectory(run, bucket, decoder, transformer, batches, 0, PER_RANK)
        if use_graph:
            assert len(run.graphs) == 3 and run.replays == STEPS - run.WARMUP
        result[name] = {"grads": [g.numpy() for g in grads], "losses": [l.numpy() for l in losses], "weights": weights.numpy()}
    q.put(result)
    dist.barrier()
    dist.destroy_process_group()


def test_forced_data_parallel_path_on_real_rccl_with_one_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29700 + os.getpid() % 2000, q))
    p.start()
    try:
        got = q.get(timeout=400)
    finally:
        p.join(timeout=120)
    assert p.exitcode == 0
    # the plain single-process step (no process group, no collectives) on the same batches
    tr, run, bucket, decoder, transformer = _make(use_graph=False)
    assert not tr.dp_active
    grads, losses, weights = _trajectory(run, bucket, decoder, transformer, _batches(), 0, PER_RANK)
    for name in ("eager", "graph"):
        for step in range(STEPS):
            assert torch.allclose(torch.as_tensor(got[name]["losses"][step]), losses[step], rtol=1e-5, atol=0), (name, step)
            assert _rel_l2(got[name]["grads"][step], grads[step]) < 1e-5, (name, step)
        assert _rel_l2(got[name]["weights"], weights) < 1e-5, name
