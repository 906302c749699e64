"""GPU parity of the whole training step (losses + gradients) against the CPU oracle's autograd."""
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _build(use_vincent, content_idx, style_idx, eps_limit):
    from style_transfer_amd.generator import Decoder, Encoder
    from style_transfer_amd.loss_network import VGG19FeatureExtractor
    from style_transfer_amd.trainer import StyleTransferStep
    from style_transfer_amd.transformer import AdaIN
    enc_sd, dec_sd = R.make_encoder_state(0), R.make_decoder_state(1)
    last = max(content_idx + style_idx)
    vgg_sd = R.make_loss_vgg_state(2, upto=last)
    enc = Encoder(weights=enc_sd).cuda().eval()
    dec = Decoder(do_train=True).cuda()
    dec.load_state_dict(dec_sd)
    ada = AdaIN(eps=eps_limit).cuda()
    vgg = VGG19FeatureExtractor([last], weights=vgg_sd).cuda()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        step = StyleTransferStep(enc, dec, ada, vgg, content_idx, style_idx, alpha=1.0, beta=10.0, gamma=1.0,
                                 delta=10.0, use_VincentStyleLoss=use_vincent)
    return step, enc_sd, dec_sd, vgg_sd


@pytest.mark.parametrize("use_vincent", [False, True])
def test_train_step_matches_oracle(use_vincent):
    content_idx, style_idx = [21], [2, 7, 14, 19, 25]        # CLI defaults (train.py:80-83)
    eps_limit = 0.1
    step, enc_sd, dec_sd, vgg_sd = _build(use_vincent, content_idx, style_idx, eps_limit)
    g = torch.Generator().manual_seed(5)
    content = torch.rand(2, 3, 32, 32, generator=g)
    style = torch.rand(2, 3, 32, 32, generator=g)

    names = ["content", "style", "variation", "hist", "total"]
    keys = ("decoder.28.weight", "decoder.28.bias", "decoder.14.weight", "decoder.1.weight", "decoder.1.bias")
    # Per-term gradients, compared in relative L2: in a deep ReLU / max-pool network with random
    # weights a 1e-6 forward difference flips an isolated ReLU gate now and then (measured: one flip
    # in 32768 activations moves ~850 gradient elements of the layer below), which a max-norm
    # comparison reports as percent-level error although every kernel matches to 1e-5 in isolation
    # (tests/test_gpu_backward.py).  The histogram
    # term sees the NORMALISED decoded image clamped to [0,1] (losses.py:170): with random decoder
    # weights ~96 % of the pixels are clamped, the gradient mask is discontinuous at the clamp
    # edges, and a 1e-6 difference in the decoded image flips individual pixels in or out; the
    # histogram kernels themselves match the oracle to 1e-5 on identical inputs (test_gpu_hist.py).
    tols = {"content": 1e-2, "style": 1e-2, "variation": 1e-2, "hist": 6e-2, "total": 6e-2}
    if use_vincent:
        tols["style"] = tols["total"] = 6e-2              # eigh backward on two different LAPACKs
    for i, name in enumerate(names):
        dec_ref = {k: v.clone().requires_grad_(True) for k, v in dec_sd.items()}
        ce = torch.tensor([1e-5], requires_grad=True)
        se = torch.tensor([1e-5], requires_grad=True)
        ref = R.train_step_losses(content, style, enc_sd, dec_ref, vgg_sd, ce, se, eps_limit, content_idx,
                                  style_idx, 1.0, 10.0, 1.0, 10.0, use_vincent)
        ref[i].backward()
        for p in list(step.decoder.parameters()) + list(step.transformer.parameters()):
            p.grad = None
        losses = step.forward_losses(content.cuda(), style.cuda())
        losses[i].backward()
        assert float(losses[i]) == pytest.approx(float(ref[i]), rel=2e-3), name
        for k in keys:
            if name == "variation" and k.endswith("28.bias"):
                continue                                     # analytically zero
            mod = step.decoder.decoder[int(k.split(".")[1])]
            got = mod.weight.grad if k.endswith("weight") else mod.bias.grad
            assert rel_l2(got, dec_ref[k].grad) < tols[name], (name, k)
        scale = max(abs(float(ce.grad)), abs(float(se.grad)), 1e-3)
        assert abs(float(step.transformer.content_eps.grad) - float(ce.grad)) < tols[name] * scale, name
        assert abs(float(step.transformer.style_eps.grad) - float(se.grad)) < tols[name] * scale, name


def test_trainer_loop_and_checkpoint(tmp_path):
    """Two epochs of Trainer.train on synthetic batches; the .pth has the reference's layout and resumes."""
    from argparse import Namespace
    from style_transfer_amd.trainer import Trainer

    class Loader(list):
        dataset = list(range(4))

    g = torch.Generator().manual_seed(3)
    batches = Loader([{"content_image": torch.rand(2, 3, 32, 32, generator=g),
                       "style_image": torch.rand(2, 3, 32, 32, generator=g)} for _ in range(2)])
    kw = dict(output_dir=str(tmp_path), num_train_epochs=2, per_device_batch_size=2, learning_rate=1e-4,
              alpha=1.0, beta=10.0, gamma=1.0, delta=10.0, eps=0.1, save_best=False, do_decoder_train=True,
              optim_name={"optim_name": "adamax", "betas": (0.9, 0.99)}, gradient_threshold=5.0,
              content_layers_idx=[21], style_layers_idx=[2, 7, 14, 19, 25], config=Namespace(note="test"),
              encoder_weights=R.make_encoder_state(0), vgg_weights=R.make_loss_vgg_state(2, upto=25))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hist = Trainer({"train": batches}, **kw).train()
    assert len(hist) == 2 and all(torch.isfinite(torch.tensor(h["total"])) for h in hist)
    ckpts = sorted(tmp_path.glob("TotalL_*_Content_*_Style_*/transformer4.pth"))
    assert len(ckpts) == 2
    ck = torch.load(ckpts[0], map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "decoder_state_dict", "optimizer_state_dict", "optim_name", "loss",
                       "completed_step", "criterion"}
    assert list(ck["model_state_dict"]) == ["content_eps", "style_eps"]
    assert list(ck["decoder_state_dict"])[0] == "decoder.1.weight" and len(ck["decoder_state_dict"]) == 18
    last = max(ckpts, key=lambda p: torch.load(p, map_location="cpu", weights_only=True)["epoch"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kw2 = dict(kw, num_train_epochs=3, resume_from_checkpoint=str(last), output_dir=str(tmp_path / "resumed"))
        hist2 = Trainer({"train": batches}, **kw2).train()
    assert [h["epoch"] for h in hist2] == [2]


def test_log_steps_rows_are_the_per_step_losses(tmp_path, capsys):
    """``--log_steps`` (Trainer(log_steps=True)): the five per-batch scalars of trainer.py:475-480, kept on the device
    during the epoch and read once at its end -- every row equals what the same step returns when it is driven
    eagerly one batch at a time (same weights trajectory: same seeds, same optimizer), and the rows are printed."""
    from style_transfer_amd.trainer import GradBucket, Trainer

    class Loader(list):
        dataset = list(range(6))

    g = torch.Generator().manual_seed(9)
    batches = Loader([{"content_image": torch.rand(2, 3, 32, 32, generator=g),
                       "style_image": torch.rand(2, 3, 32, 32, generator=g)} for _ in range(3)])
    kw = dict(output_dir=str(tmp_path), num_train_epochs=1, per_device_batch_size=2, learning_rate=1e-3,
              alpha=1.0, beta=10.0, gamma=1.0, delta=10.0, eps=0.1, save_best=False, do_decoder_train=True,
              optim_name={"optim_name": "adam"}, content_layers_idx=[21], style_layers_idx=[2, 7, 14, 19, 25],
              encoder_weights=R.make_encoder_state(0), vgg_weights=R.make_loss_vgg_state(2, upto=25),
              decoder_weights={k.removeprefix("decoder."): v for k, v in R.make_decoder_state(1).items()},
              use_pretrained_WCTDECODER=True, step_frequency=10.0)       # (no LR step inside the epoch)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hist = Trainer({"train": batches}, log_steps=True, **kw).train()
        rows = hist[0]["steps"]
        assert len(rows) == 3 and all(len(r) == 5 for r in rows)
        printed = capsys.readouterr().out
        assert "step 0: content" in printed and "step 2: content" in printed
        # the same three steps, driven by hand
        tr = Trainer({"train": batches}, **dict(kw, output_dir=str(tmp_path / "by_hand")))
        encoder, decoder, transformer, loss_vgg = tr.build_model()
        params = tr.trainable_parameters(transformer, decoder)
        opt = tr.get_optimizer("adam", params, 1e-3, {"optim_name": "adam"})
        step, bucket = tr.make_step(encoder, decoder, transformer, loss_vgg), GradBucket(params)
        for row, b in zip(rows, batches):
            out = tr.optimize_step(step, opt, bucket, b["content_image"].cuda(), b["style_image"].cuda())
            for got, want in zip(row, out):
                assert got == pytest.approx(float(want), rel=1e-5), (row, [float(v) for v in out])
    # the epoch means the trainer reports are the means of those rows
    for i, k in enumerate(("content", "style", "variation", "hist", "total")):
        assert hist[0][k] == pytest.approx(sum(r[i] for r in rows) / 3, rel=1e-5)


def _trainer_for_steps(use_graph):
    import warnings
    from style_transfer_amd.trainer import GradBucket, GraphedStep, Trainer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr = Trainer({"train": []}, output_dir="/tmp/st_graph_test", num_train_epochs=1, per_device_batch_size=2,
                     learning_rate=1e-3, alpha=1.0, beta=10.0, gamma=1.0, delta=10.0, eps=0.1, do_decoder_train=True,
                     optim_name={"optim_name": "adam"}, content_layers_idx=[21], style_layers_idx=[2, 7, 14, 19, 25],
                     encoder_weights=R.make_encoder_state(0), vgg_weights=R.make_loss_vgg_state(2, upto=25),
                     device="cuda", use_hip_graph=True)
        assert tr.graph_capable()
        encoder, decoder, transformer, loss_vgg = tr.build_model()
        decoder.load_state_dict(R.make_decoder_state(1))
        params = tr.trainable_parameters(transformer, decoder)
        lr = torch.tensor(1e-3, device="cuda")
        opt = tr.get_optimizer("adam", params, lr, tr.graph_optim_kwargs())     # capturable Adam in both runs
        step = tr.make_step(encoder, decoder, transformer, loss_vgg)
    bucket = GradBucket(params)
    run = GraphedStep(tr, step, opt, bucket) if use_graph else \
        (lambda c, s: tr.optimize_step(step, opt, bucket, c, s))
    return run, decoder, transformer, lr


def test_hip_graph_replay_matches_eager_steps():
    """GraphedStep (3 eager warm-up steps, capture, replays) walks the same trajectory as eager steps: same losses
    and same decoder weights after 8 optimizer steps on changing batches, including a learning-rate change that
    has to reach the captured optimizer through the device-resident lr tensor, and a ragged batch run eagerly."""
    g = torch.Generator().manual_seed(9)
    batches = [(torch.rand(2, 3, 64, 64, generator=g).cuda(), torch.rand(2, 3, 64, 64, generator=g).cuda())
               for _ in range(8)]
    batches.insert(6, (torch.rand(1, 3, 64, 64, generator=g).cuda(), torch.rand(1, 3, 64, 64, generator=g).cuda()))
    results = []
    for use_graph in (False, True):
        run, decoder, transformer, lr = _trainer_for_steps(use_graph)
        totals = []
        for i, (c, s) in enumerate(batches):
            if i == 5:
                lr.fill_(3e-4)
            losses = run(c, s)
            totals.append(float(losses[4].detach()))
        if use_graph:
            assert run.graph is not None and run.shape == ((2, 3, 64, 64), (2, 3, 64, 64))
        torch.cuda.synchronize()
        results.append((totals, [p.detach().clone() for p in decoder.parameters()],
                        [p.detach().clone() for p in transformer.parameters()]))
    (t0, d0, a0), (t1, d1, a1) = results
    for a, b in zip(t0, t1):
        assert abs(a - b) <= 1e-5 * abs(a), (t0, t1)
    for a, b in zip(d0 + a0, d1 + a1):
        assert rel_l2(b, a) < 1e-5


def test_trainer_loop_with_hip_graph_matches_eager(tmp_path, monkeypatch):
    """Trainer.train with use_hip_graph=True (GraphedStep inside the real loop: cosine LR schedule stepping the
    device-resident learning rate, gradient clamp, checkpoint written with the capturable optimizer, resume) gives
    the same epoch losses as the same loop with the capture switched off (WARMUP never reached: every step eager,
    same capturable optimizer arithmetic)."""
    from style_transfer_amd.trainer import Trainer, GraphedStep

    class Loader(list):
        dataset = list(range(12))

    g = torch.Generator().manual_seed(4)
    batches = Loader([{"content_image": torch.rand(2, 3, 48, 48, generator=g),
                       "style_image": torch.rand(2, 3, 48, 48, generator=g)} for _ in range(6)])
    import warnings
    hists = []
    for graph in (False, True):
        kw = dict(output_dir=str(tmp_path / f"g{int(graph)}"), num_train_epochs=2, per_device_batch_size=2,
                  learning_rate=1e-3, alpha=1.0, beta=10.0, gamma=1.0, delta=10.0, eps=0.1, save_best=False,
                  do_decoder_train=True, optim_name={"optim_name": "adam"}, gradient_threshold=5.0,
                  content_layers_idx=[21], style_layers_idx=[2, 7, 14, 19, 25],
                  encoder_weights=R.make_encoder_state(0),
                  decoder_weights={k.removeprefix("decoder."): v for k, v in R.make_decoder_state(1).items()},
                  vgg_weights=R.make_loss_vgg_state(2, upto=25), use_pretrained_WCTDECODER=True, use_hip_graph=True)
        monkeypatch.setattr(GraphedStep, "WARMUP", 3 if graph else 10 ** 9)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            hists.append(Trainer({"train": batches}, **kw).train())
    eager, graphed = hists
    assert len(graphed) == 2
    for a, b in zip(eager, graphed):
        for k in ("content", "style", "variation", "hist", "total"):
            assert abs(a[k] - b[k]) <= 1e-5 * max(abs(a[k]), 1e-6), (k, a[k], b[k])
    ckpts = sorted((tmp_path / "g1").glob("TotalL_*/transformer12.pth"))
    assert ckpts
    last = max(ckpts, key=lambda p: torch.load(p, map_location="cpu", weights_only=True)["epoch"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkeypatch.setattr(GraphedStep, "WARMUP", 3)
        kw2 = dict(kw, num_train_epochs=3, resume_from_checkpoint=str(last), output_dir=str(tmp_path / "resumed"))
        hist2 = Trainer({"train": batches}, **kw2).train()
    assert [h["epoch"] for h in hist2] == [2] and hist2[0]["total"] == hist2[0]["total"]


def test_eager_forward_after_graph_replays_uses_current_weights():
    """a HIP-graph replay moves the decoder weights in place through captured optimizer kernels without
    bumping Parameter._version, so a packed-weight cache keyed on the version alone would serve STALE packed copies
    to an eager forward between replays (per-epoch eval, Trainer.stylize after train()).  After every batch of
    replays an eager stylize must equal a fresh Decoder holding the current state_dict."""
    from style_transfer_amd.generator import Decoder
    from style_transfer_amd.trainer import Trainer
    run, decoder, transformer, lr = _trainer_for_steps(True)
    lr.fill_(1e-2)                                     # large steps: stale weights would be far off
    encoder = run.step.encoder
    g = torch.Generator().manual_seed(31)
    batches = [(torch.rand(2, 3, 64, 64, generator=g).cuda(), torch.rand(2, 3, 64, 64, generator=g).cuda())
               for _ in range(10)]
    c, s = batches[0]
    seen = []
    for i, (bc, bs) in enumerate(batches):
        run(bc, bs)
        if i in (4, 6, 9):                               # after 2, 4 and 7 replays (the capture is call 3)
            assert run.graph is not None and run.replays == i - 2
            got = Trainer.stylize(encoder, decoder, transformer, c, s, alpha=1.0, normalize=False)
            fresh = Decoder().cuda().eval()
            fresh.load_state_dict(decoder.state_dict())
            want = Trainer.stylize(encoder, fresh, transformer, c, s, alpha=1.0, normalize=False)
            assert rel_l2(got, want) < 1e-6, i
            seen.append(got)
    assert rel_l2(seen[1], seen[0]) > 1e-3 and rel_l2(seen[2], seen[1]) > 1e-3     # the weights did move


@pytest.mark.parametrize("eps_limit,moves", [(1e-5, False), (0.1, True)])
def test_clamped_eps_parameters_are_not_touched_by_the_optimizer(eps_limit, moves):
    """transformer.py:27-36: in a clamped branch adjust_sigmoid returns a Python float, the eps parameter gets NO
    gradient and torch's optimizers skip it -- no update, no weight decay.  With the CLI default --eps 1e-5 the
    parameter (init 1e-5, sqrt = 3.2e-3 > 1e-5) is always clamped.  Here the clamp lives on the device and delivers
    a zero gradient, so the trainer restores clamped parameters after optimizer.step(): with AdamW and a large
    weight decay they must stay bit-identical; with eps_limit 0.1 (unclamped) they must move."""
    import warnings
    from style_transfer_amd.trainer import GradBucket, Trainer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr = Trainer({"train": []}, output_dir="/tmp/st_eps_test", num_train_epochs=1, per_device_batch_size=2,
                     learning_rate=1e-2, alpha=1.0, beta=10.0, gamma=1.0, delta=10.0, eps=eps_limit, do_decoder_train=False,
                     optim_name={"optim_name": "adamw", "weight_decay": 0.5}, content_layers_idx=[21],
                     style_layers_idx=[2, 7, 14, 19, 25], encoder_weights=R.make_encoder_state(0),
                     vgg_weights=R.make_loss_vgg_state(2, upto=25), device="cuda")
        encoder, decoder, transformer, loss_vgg = tr.build_model()
        decoder.load_state_dict(R.make_decoder_state(1))
        params = tr.trainable_parameters(transformer, decoder)
        opt = tr.get_optimizer("adamw", params, 1e-2, tr.graph_optim_kwargs())
        step = tr.make_step(encoder, decoder, transformer, loss_vgg)
    bucket = GradBucket(params)
    g = torch.Generator().manual_seed(12)
    before = [p.detach().clone() for p in transformer.parameters()]
    for _ in range(3):
        tr.optimize_step(step, opt, bucket, torch.rand(2, 3, 32, 32, generator=g).cuda(),
                         torch.rand(2, 3, 32, 32, generator=g).cuda())
    after = [p.detach().clone() for p in transformer.parameters()]
    if moves:
        assert all(not torch.equal(a, b) for a, b in zip(after, before))
    else:
        assert all(torch.equal(a, b) for a, b in zip(after, before))


def test_gram_step_with_lookahead_walks_the_same_trajectory():
    """optimize_step(prefetch=next batch) with VincentStyleLoss: the next batch's weight-independent forward (Normalize,
    loss-VGG passes over the originals, frozen encoder, cov_stl^(1/2)) runs on a side stream under this batch's
    eigen-decompositions.  Same kernels on the same data: losses and weights must equal the plain sequence."""
    import warnings
    from style_transfer_amd.trainer import GradBucket, Trainer

    def run(lookahead):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tr = Trainer({"train": []}, output_dir="/tmp/st_la_test", num_train_epochs=1, per_device_batch_size=2,
                         learning_rate=1e-3, alpha=1.0, beta=10.0, gamma=1.0, delta=10.0, eps=0.1, do_decoder_train=True,
                         optim_name={"optim_name": "adam"}, use_VincentStyleLoss=True, content_layers_idx=[21],
                         style_layers_idx=[2, 7, 14, 19, 25], encoder_weights=R.make_encoder_state(0),
                         vgg_weights=R.make_loss_vgg_state(2, upto=25), device="cuda")
            encoder, decoder, transformer, loss_vgg = tr.build_model()
            decoder.load_state_dict(R.make_decoder_state(1))
            params = tr.trainable_parameters(transformer, decoder)
            opt = tr.get_optimizer("adam", params, 1e-3, {})
            step = tr.make_step(encoder, decoder, transformer, loss_vgg)
        bucket = GradBucket(params)
        g = torch.Generator().manual_seed(77)
        batches = [(torch.rand(2, 3, 64, 64, generator=g).cuda(), torch.rand(2, 3, 64, 64, generator=g).cuda())
                   for _ in range(5)]
        out = []
        for i, (c, s) in enumerate(batches):
            nxt = batches[i + 1] if lookahead and i + 1 < len(batches) else None
            losses = tr.optimize_step(step, opt, bucket, c, s, prefetch=nxt)
            if lookahead and nxt is not None:
                assert tr._front_next is not None and tr._front_next[0] is nxt[0]       # the prefetch was issued
            out.append(torch.stack([l.detach().reshape(()) for l in losses]).cpu())
        torch.cuda.synchronize()
        assert tr._front_next is None
        w = torch.cat([p.detach().flatten() for p in list(decoder.parameters()) + list(transformer.parameters())]).cpu()
        return torch.stack(out), w
    l0, w0 = run(False)
    l1, w1 = run(True)
    torch.testing.assert_close(l1, l0, rtol=1e-6, atol=0)
    torch.testing.assert_close(w1, w0, rtol=1e-6, atol=1e-9)
    # a prefetched front that does not belong to the batch handed in is dropped, not used
    l2, _ = run(False)
    torch.testing.assert_close(l2, l0, rtol=0, atol=0)
