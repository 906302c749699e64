"""Training step, loss mix, checkpoint save/resume and data-parallel glue with the reference's
``Trainer`` surface (src/models/trainer.py), every tensor op of the step on HIP kernels.

Differences that do not change results: the encoder and the loss network are
frozen instead of accumulating never-read gradients; the decoded image goes through the loss
VGG once for both extractors; the stylised histogram is computed once for both Hellinger terms;
per-step scalars stay on the device and are read once per epoch.

Data parallel (one process per GPU, ``torch.distributed`` backend "nccl" = RCCL over xGMI): the
batch is sharded by rank, one flat fp32 bucket (decoder grads + the two AdaIN eps grads, 14.02 MB)
is all-reduced per step, the Hellinger scalar under the outer sqrt is all-reduced in the forward so
the histogram loss equals the single-GPU large-batch value, the AdaIN gradient clamp is applied
after the reduction, and only rank 0 writes checkpoints.
"""
from __future__ import annotations

import datetime
import math
import os
import warnings
from argparse import Namespace
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.optim as optim
from torch.optim import lr_scheduler

from .generator import Decoder, Encoder
from .loss_network import IMAGENET_MEAN, IMAGENET_STD, Normalize, VGG19FeatureExtractor
from .losses import AdaINContentLoss, AdaINStyleLoss, HistLoss, TVLoss, VincentStyleLoss
from .transformer import AdaIN

PRJ_NAME = "Style_transfer"   # trainer.py:36


# ---------------------------------------------------------------------------------------------
# data-parallel helpers
# ---------------------------------------------------------------------------------------------
def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


class _AllReduceSumScaledGrad(torch.autograd.Function):
    """Q = sum over ranks of q (forward); d q_local = world_size * dQ (backward).

    With the per-rank gradients averaged afterwards (mean all-reduce of the bucket) the scaled
    backward makes the histogram term's gradient equal d L_global / d theta exactly
    (the sqrt of HistLoss sits outside the batch sum)."""

    @staticmethod
    def forward(ctx, q):
        dist = _dist()
        out = q.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        ctx.world = dist.get_world_size()
        return out

    @staticmethod
    def backward(ctx, g):
        return g * ctx.world


def allreduce_q(q):
    return _AllReduceSumScaledGrad.apply(q) if _dist() is not None else q


class _InjectReducedQ(torch.autograd.Function):
    """The same exchange cut in two, for a step replayed from HIP graphs: the all-reduce itself runs eagerly
    BETWEEN two graphs on a static buffer ``q_sum``; this node (first thing of the second graph) hands that
    buffer's value to the loss and sends ``world_size * dQ`` back to the local q."""

    @staticmethod
    def forward(ctx, q_local, q_sum, world):
        ctx.world = world
        return q_sum.clone()

    @staticmethod
    def backward(ctx, g):
        return g * ctx.world, None, None


class GradBucket:
    """ONE flat fp32 buffer holds every trainable gradient: each parameter's ``.grad`` is a view into it
    (no pack / unpack copies), the data-parallel exchange is a single in-place RCCL all-reduce of the buffer
    (decoder + 2 AdaIN eps = 14.02 MB), zeroing it is one memset."""

    def __init__(self, params: List[nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat: Optional[torch.Tensor] = None
        self._views: List[torch.Tensor] = []

    def attach(self) -> None:
        """(Re-)point every ``.grad`` at its slice of the flat buffer.  Cheap when nothing changed (a pointer
        comparison per parameter); a ``zero_grad(set_to_none=True)`` by a caller is repaired here."""
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
            self._views, off = [], 0
            for p in self.params:
                self._views.append(self.flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
        for p, v in zip(self.params, self._views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def zero(self) -> None:
        self.attach()
        self.flat.zero_()

    def all_reduce_sum(self) -> None:
        dist = _dist()
        if dist is not None:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)

    def reduce_mean(self) -> None:
        """Stand-alone mean all-reduce of whatever gradients the parameters hold (the training step uses
        ``zero`` / ``all_reduce_sum`` and folds the 1 / world_size into its update phase)."""
        dist = _dist()
        if dist is None or dist.get_world_size() == 1:
            return
        held = [p.grad for p in self.params]
        self.attach()
        for g, v in zip(held, self._views):
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                v.copy_(g)
        self.all_reduce_sum()
        self.flat.div_(dist.get_world_size())


# ---------------------------------------------------------------------------------------------
# the step engine (hot path)
# ---------------------------------------------------------------------------------------------
class StyleTransferStep:
    """Forward half of the train-step body (trainer.py:426-464) + compute_loss (:246-279).

    The loss is produced in two stages so that a data-parallel step has exactly one place where ranks talk in
    the forward: ``forward_partials`` (everything up to the per-rank scalars q under HistLoss's outer sqrt) and
    ``combine`` (the rest, given the batch-wide q)."""

    def __init__(self, encoder: Encoder, decoder: Decoder, transformer: AdaIN, loss_vgg: VGG19FeatureExtractor,
                 content_layers_idx: List[int], style_layers_idx: List[int], alpha: float, beta: float, gamma: float,
                 delta: float, use_VincentStyleLoss: bool = False):
        self.encoder, self.decoder, self.transformer, self.loss_vgg = encoder, decoder, transformer, loss_vgg
        self.content_layers_idx, self.style_layers_idx = list(content_layers_idx), list(style_layers_idx)
        self.alpha, self.beta, self.gamma, self.delta = alpha, beta, gamma, delta
        self.variation_loss = TVLoss()
        if use_VincentStyleLoss:
            self.style_loss = VincentStyleLoss()
        else:
            self.style_loss = AdaINStyleLoss()
        self.content_loss = AdaINContentLoss()
        self.hist_loss = HistLoss()
        self.norm = Normalize(IMAGENET_MEAN, IMAGENET_STD)

    def partial_losses(self, content_features, style_features, stylized_outputs, style_imgs, content_imgs,
                       prepared_style=None, on_style_jobs_issued=None):
        """First stage of Trainer.compute_loss: (content, style, variation, q[2]) where q = the local sums under
        the sqrt of the two Hellinger terms (stylised vs style, stylised vs content).  ``prepared_style``: the
        style-image side of VincentStyleLoss computed ahead (``prepare_targets``, side streams);
        ``on_style_jobs_issued``: called once its stylised-side jobs are on their streams (see forward_multi)."""
        loss_content = sum(self.content_loss(f['model_output'], f['orginal']) for f in content_features)
        if prepared_style is not None:
            loss_style = self.style_loss.forward_multi([f['model_output'] for f in style_features], prepared_style,
                                                       on_jobs_issued=on_style_jobs_issued)
        else:
            loss_style = sum(self.style_loss(f['model_output'], f['orginal']) for f in style_features)
        variation_loss = self.variation_loss(stylized_outputs)
        blk = self.hist_loss.histogram_block
        stylized_hist = blk(stylized_outputs)             # computed once, used by both terms
        with torch.no_grad():
            style_hist, content_hist = blk(style_imgs), blk(content_imgs)
        q = torch.stack([self.hist_loss.q_from_hists(stylized_hist, style_hist),
                         self.hist_loss.q_from_hists(stylized_hist, content_hist)])
        return loss_content, loss_style, variation_loss, q

    def combine(self, partials, q_sum, batch: int):
        """Second stage: the five scalars of Trainer.compute_loss (trainer.py:269-279)."""
        loss_content, loss_style, variation_loss, _ = partials
        histogram_loss = 0.4 * self.hist_loss.from_q(q_sum[0], batch) + 0.6 * self.hist_loss.from_q(q_sum[1], batch)
        total_loss = self.alpha * loss_content + self.beta * loss_style + self.gamma * variation_loss \
            + self.delta * histogram_loss
        return loss_content, loss_style, variation_loss, histogram_loss, total_loss

    def compute_loss(self, content_features, style_features, stylized_outputs, style_imgs, content_imgs,
                     global_batch: Optional[int] = None):
        """Trainer.compute_loss: *_features are lists of {"model_output", "orginal"} dicts."""
        partials = self.partial_losses(content_features, style_features, stylized_outputs, style_imgs, content_imgs)
        batch = stylized_outputs.shape[0] if global_batch is None else global_batch
        return self.combine(partials, allreduce_q(partials[3]), batch)

    # The forward in two halves.  ``front``: everything that does not depend on a TRAINED weight -- Normalize of the
    # originals, their loss-VGG passes, the frozen encoder on content and style, the style-image side of
    # VincentStyleLoss.  ``rest``: AdaIN (trainable eps), decoder, the stylised image's loss-VGG pass, the losses.
    # The split is what lets a Gram-loss step compute the NEXT batch's front while this batch's eigen-decompositions
    # (a 13 ms chain of tiny launches that leaves the chip idle) are running: Trainer.optimize_step(prefetch=...).
    def front(self, content_imgs: torch.Tensor, style_imgs: torch.Tensor, inline_targets: bool = False) -> dict:
        # The style image's loss-VGG pass goes FIRST: with VincentStyleLoss everything that depends on the style
        # image alone (covariances, cov_stl^(1/2)) is then issued on side streams and overlaps the encoder
        # convolutions below.  Same values as the reference's order (trainer.py:453).
        with torch.no_grad():
            style_n = self.norm(style_imgs)
            org_style = self.loss_vgg.forward_multi(style_n, [self.style_layers_idx])[0]
            prepared_style = self.style_loss.prepare_targets(org_style, inline=inline_targets) \
                if hasattr(self.style_loss, "prepare_targets") else None
            # frozen encoder: forward only
            if content_imgs.shape == style_imgs.shape:
                # one pass over [content; style]: same kernels, same per-image results (no cross-image op in
                # the encoder), twice the grid -- small training batches fill the chip better
                n = content_imgs.shape[0]
                feats = self.encoder(torch.cat([content_imgs, style_imgs]))
                encode_Cfeatures, encode_Sfeatures = feats[:n], feats[n:]
            else:
                encode_Cfeatures = self.encoder(content_imgs)
                encode_Sfeatures = self.encoder(style_imgs)
            content_n = self.norm(content_imgs)
            org_content = self.loss_vgg.forward_multi(content_n, [self.content_layers_idx])[0]
        return {"style_n": style_n, "content_n": content_n, "org_style": org_style, "org_content": org_content,
                "prepared_style": prepared_style, "encode_C": encode_Cfeatures, "encode_S": encode_Sfeatures}

    def rest(self, fr: dict, on_style_jobs_issued=None):
        transformed_features = self.transformer(fr["encode_C"], fr["encode_S"], alpha=1.0)   # trainer.py:433-434
        decode_imgs = self.decoder(transformed_features)
        decode_n = self.norm(decode_imgs)
        gen_content, gen_style = self.loss_vgg.forward_multi(decode_n, [self.content_layers_idx, self.style_layers_idx])
        # the reference indexes the getter's layer-ordered outputs by position (trainer.py:448-457)
        content_features = [{"orginal": o, "model_output": g} for o, g in zip(fr["org_content"], gen_content)]
        style_features = [{"orginal": o, "model_output": g} for o, g in zip(fr["org_style"], gen_style)]
        return self.partial_losses(content_features, style_features, decode_n, fr["style_n"], fr["content_n"],
                                   fr["prepared_style"], on_style_jobs_issued)

    def forward_partials(self, content_imgs: torch.Tensor, style_imgs: torch.Tensor, front: Optional[dict] = None,
                         on_style_jobs_issued=None):
        return self.rest(front if front is not None else self.front(content_imgs, style_imgs), on_style_jobs_issued)

    def forward_losses(self, content_imgs: torch.Tensor, style_imgs: torch.Tensor, global_batch: Optional[int] = None):
        partials = self.forward_partials(content_imgs, style_imgs)
        batch = content_imgs.shape[0] if global_batch is None else global_batch
        return self.combine(partials, allreduce_q(partials[3]), batch)


# ---------------------------------------------------------------------------------------------
class GraphedStep:
    """One training step captured in HIP graphs (``torch.cuda.CUDAGraph`` = hipGraph on ROCm) and replayed.

    A step is ~400 kernel launches through ctypes plus the autograd tape; at small shapes (or on a busy host)
    the GPU waits for the host.  Every launch of the step goes to torch's current stream, the step has no
    host read in it (AdaIN's eps stays on the device), workspaces come from torch's allocator -- so it records
    into graphs and a replay costs ~0.1 ms of host time.

    Single process: ONE graph holds the whole optimize_step.  Data parallel (world_size > 1): the step has two
    exchanges (the 2-float Hellinger sums in the forward, the flat gradient bucket after the backward), so it
    is THREE graphs sharing one memory pool with the two eager all-reduces between them::

        graph A  zero the bucket, forward up to the local q        (Trainer.step_forward)
        eager    all-reduce(q_sum[2])
        graph B  rest of the loss, backward into the bucket        (Trainer.step_backward)
        eager    all-reduce(bucket)
        graph C  1/world, AdaIN grad clamp, optimizer.step()       (Trainer.step_update)

    The autograd tape built while graph A records lives on the host only during the capture; graph B records the
    backward kernels it drives, reading the activations graph A left in the shared pool.

    Inputs are copied into static buffers; the returned losses are static tensors overwritten by every replay.
    Needs an optimizer built with ``capturable=True`` (Trainer does that when ``use_hip_graph`` is set) and a
    fixed batch shape: other shapes run eagerly."""

    WARMUP = 3

    def __init__(self, trainer: "Trainer", step: StyleTransferStep, optimizer, bucket: GradBucket):
        self.trainer, self.step, self.optimizer, self.bucket = trainer, step, optimizer, bucket
        self.graph = None
        self.graphs = []
        self.shape = None
        self.static_content = self.static_style = None
        self.losses = None
        self.q_sum = None
        self.eager_calls = 0
        self.replays = 0
        self._side = None

    def _eager(self, content, style):
        return self.trainer.optimize_step(self.step, self.optimizer, self.bucket, content, style)

    def _capture(self):
        tr, dist = self.trainer, _dist()
        torch.cuda.synchronize()
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # side-stream warm-up: intended
        self.bucket.attach()
        if not tr.dp_active:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.losses = self._eager(self.static_content, self.static_style)
            self.graphs = [graph]
        else:
            ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # thread-local capture mode: the process group's watchdog thread may query events while we record
            mode = dict(capture_error_mode="thread_local")
            with torch.cuda.graph(ga, **mode):
                partials = tr.step_forward(self.step, self.bucket, self.static_content, self.static_style)
                self.q_sum = partials[3].detach().clone()
            dist.all_reduce(self.q_sum, op=dist.ReduceOp.SUM)     # same collective sequence as every later step
            with torch.cuda.graph(gb, pool=ga.pool(), **mode):
                self.losses = tr.step_backward(self.step, partials, self.q_sum,
                                               self.static_content.shape[0] * tr.world_size)
            del partials
            self.bucket.all_reduce_sum()
            with torch.cuda.graph(gc, pool=ga.pool(), **mode):
                tr.step_update(self.step, self.optimizer, self.bucket)
            self.graphs = [ga, gb, gc]
        self.graph = self.graphs[0]
        torch.cuda.synchronize()

    def _replay(self):
        if len(self.graphs) == 1:
            self.graphs[0].replay()
        else:
            dist = _dist()
            ga, gb, gc = self.graphs
            ga.replay()
            dist.all_reduce(self.q_sum, op=dist.ReduceOp.SUM)
            gb.replay()
            self.bucket.all_reduce_sum()
            gc.replay()
        self.replays += 1
        from .generator import _PackedWeightCache
        _PackedWeightCache.invalidate_all()        # weights moved without a ._version bump

    def __call__(self, content: torch.Tensor, style: torch.Tensor):
        key = (tuple(content.shape), tuple(style.shape))
        if self.shape is not None and key != self.shape:
            return self._eager(content, style)            # ragged last batch: eager
        if self.graph is None:
            if self.eager_calls < self.WARMUP:             # allocator, packed-weight caches, optimizer state
                # Warm-up steps run on a side stream (the PyTorch capture recipe): autograd's AccumulateGrad
                # nodes remember the stream they were created on, and nodes tied to the legacy default stream
                # make the backward inside the capture synchronise with it, which breaks the capture.
                self.eager_calls += 1
                if self._side is None:
                    self._side = torch.cuda.Stream()
                cur = torch.cuda.current_stream()
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    out = self._eager(content, style)
                cur.wait_stream(self._side)
                for t in (content, style, *out):
                    t.record_stream(self._side)
                return out
            self.shape = key
            self.static_content, self.static_style = content.clone(), style.clone()
            self._capture()
            # the capture recorded the step without running it: the replay below IS this call's step
        else:
            self.static_content.copy_(content, non_blocking=True)
            self.static_style.copy_(style, non_blocking=True)
        self._replay()
        return self.losses


class Trainer:
    """Constructor keywords, optimizer names, checkpoint layout and resume rules of the reference
    Trainer (trainer.py:39-108, 201-232, 304-388, 607-655).  Extra keywords (all optional):
    ``encoder_weights`` / ``decoder_weights`` / ``vgg_weights`` (paths or state dicts; the blobs
    the reference hard-codes are not shipped), ``device``, ``compute_dtype`` ("f32" | "bf16" | "bf16x3"),
    ``use_hip_graph`` (capture the training step in a HIP graph and replay it, GraphedStep)."""

    def __init__(self, dataloaders, output_dir: str, num_train_epochs: int, per_device_batch_size: int,
                 learning_rate: float, alpha: float, beta: float, gamma: float, seed: int = 42,
                 do_eval_per_epoch: bool = True, plot_per_epoch: bool = False, save_best: bool = True,
                 resume_from_checkpoint: str = None, vgg_model_type: str = '19',
                 optim_name: dict = {'optim_name': 'adam'}, with_tracking: bool = False,
                 log_weights_cpkt: bool = False, do_decoder_train: bool = False,
                 use_pretrained_WCTDECODER: bool = False, delta: float = 2, eps: float = 1e-5,
                 step_frequency: float = 0.5, gradient_threshold: float = None, config: Namespace = None,
                 grayscale_content_transform: bool = False, use_VincentStyleLoss: bool = False,
                 content_layers_idx: List[int] = [12, 16, 21], style_layers_idx: List[int] = [0, 5, 10, 19, 28],
                 encoder_weights=None, decoder_weights=None, vgg_weights=None, device=None,
                 compute_dtype: str = "f32", use_hip_graph: bool = False, log_steps: bool = False,
                 gpu_augment=None):
        self.output_dir = output_dir
        # (content, style) GpuAugment pair: the post-ToTensor half of the reference's transform_content /
        # transform_style (train.py:116-135) applied per batch on the device; None = the loader's transforms did it
        self.gpu_augment = gpu_augment
        self.dataloaders = dataloaders.__call__() if callable(dataloaders) else dataloaders
        self.per_device_batch_size = per_device_batch_size
        self.alpha, self.beta, self.gamma, self.delta, self.eps = alpha, beta, gamma, delta, eps
        self.vgg_model_type = vgg_model_type
        self.learning_rate = learning_rate
        self.content_layers_idx = list(content_layers_idx)
        self.style_layers_idx = list(style_layers_idx)
        self.num_train_epochs = num_train_epochs
        self.step_frequency = step_frequency
        self.optim_name = dict(optim_name)
        self.gradient_threshold = gradient_threshold
        self.do_decoder_train = do_decoder_train
        self.use_pretrained_WCTDECODER = use_pretrained_WCTDECODER
        self.grayscale_content_transform = grayscale_content_transform
        self.use_VincentStyleLoss = use_VincentStyleLoss
        self.with_tracking = with_tracking
        self.log_weights_cpkt = log_weights_cpkt
        self.plot_per_epoch = plot_per_epoch
        self.save_best = save_best
        self.resume_from_checkpoint = resume_from_checkpoint
        self.do_eval_per_epoch = do_eval_per_epoch
        self.config = config if config is not None else Namespace()
        self.seed = seed
        self.device = torch.device(device) if device is not None else \
            torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.encoder_weights, self.decoder_weights, self.vgg_weights = encoder_weights, decoder_weights, vgg_weights
        self.compute_dtype = compute_dtype       # "f32" | "bf16" | "bf16x3" (precision.py)
        self.use_hip_graph = bool(use_hip_graph)  # replay the step from a HIP graph (GraphedStep)
        # the five per-batch scalars of trainer.py:475-480 (content, style, variation, hist, total): kept in a device
        # tensor during the epoch and read ONCE at its end -- the reference's five .item() calls per step would put a
        # host synchronisation into every step
        self.log_steps = bool(log_steps)
        if self.with_tracking:
            raise RuntimeError("wandb tracking is outside the hot path this build covers (SURVEY.md section 2)")
        self.wandb = None
        dist = _dist()
        self.rank = dist.get_rank() if dist is not None else 0
        self.world_size = dist.get_world_size() if dist is not None else 1
        # ST_FORCE_DP=1 with an initialised process group: take the data-parallel code path (both all-reduces, the
        # three-graph replay) even with ONE rank -- lets a 1-GPU box drive the real RCCL backend through it
        self.dp_active = self.world_size > 1 or (dist is not None and bool(os.environ.get("ST_FORCE_DP")))

    # ---- factories ------------------------------------------------------------------------
    @staticmethod
    def get_feature_extractor(selected_indices, vgg_model_type="19", device='cpu', weights=None):
        from .loss_network import get_feature_extractor
        return get_feature_extractor(selected_indices, vgg_model_type, device, weights)

    @staticmethod
    def get_optimizer(optimizer_name: str, parameters, learning_rate: float = 5e-5, kwargs: dict = None):
        """trainer.py:201-232: ten torch optimizers by (case-insensitive) name; stock torch objects."""
        optimizer_name = optimizer_name.lower()
        kwargs_clone = dict(kwargs or {})
        kwargs_clone.pop('optim_name', None)
        table = {'sgd': optim.SGD, 'adam': optim.Adam, 'rmsprop': optim.RMSprop, 'adagrad': optim.Adagrad,
                 'adadelta': optim.Adadelta, 'adamw': optim.AdamW, 'adamax': optim.Adamax,
                 'sparseadam': optim.SparseAdam, 'lbfgs': optim.LBFGS, 'rprop': optim.Rprop}
        if optimizer_name not in table:
            raise ValueError(f"Optimizer '{optimizer_name}' not supported.")
        try:
            return table[optimizer_name](parameters, lr=learning_rate, **kwargs_clone)
        except TypeError:
            raise TypeError("Invalid argument for optimizer!")

    @staticmethod
    def _capturable(optimizer_name: str) -> bool:
        """Does this torch optimizer take ``capturable=True`` (state and step count on the device)?"""
        import inspect
        cls = {'sgd': optim.SGD, 'adam': optim.Adam, 'rmsprop': optim.RMSprop, 'adagrad': optim.Adagrad,
               'adadelta': optim.Adadelta, 'adamw': optim.AdamW, 'adamax': optim.Adamax, 'rprop': optim.Rprop
               }.get(optimizer_name.lower())
        return cls is not None and 'capturable' in inspect.signature(cls.__init__).parameters

    def graph_capable(self) -> bool:
        """HIP-graph replay of the step: asked for, a capturable torch optimizer and no rocSOLVER
        eigendecomposition (VincentStyleLoss) in the step.  Data parallel included (GraphedStep: three graphs
        around the two eager all-reduces)."""
        return (self.use_hip_graph and not self.use_VincentStyleLoss
                and self.device.type == 'cuda' and self._capturable(self.optim_name['optim_name']))

    def graph_optim_kwargs(self) -> dict:
        """Optimizer kwargs for graph replay: ``capturable=True`` (optimizer state and step count on the device)."""
        kw = dict(self.optim_name)
        if self.graph_capable():
            kw['capturable'] = True
        return kw

    def build_model(self):
        """trainer.py:234-244 (one loss network serves both index lists)."""
        encoder = Encoder(weights=self.encoder_weights).eval().to(self.device)
        use_wct = self.use_pretrained_WCTDECODER if self.resume_from_checkpoint is None else False
        decoder = Decoder(use_pretrained_WCT=use_wct, do_train=self.do_decoder_train,
                          weights=self.decoder_weights).to(self.device)
        transformer = AdaIN(eps=self.eps).to(self.device)
        last = max(self.content_layers_idx + self.style_layers_idx)
        loss_vgg = VGG19FeatureExtractor([last], weights=self.vgg_weights).to(self.device)
        return encoder, decoder, transformer, loss_vgg

    # ---- checkpoint -----------------------------------------------------------------------
    def checkpoint_dict(self, transformer, transformer_optimizer, decoder, info):
        """Exactly the dict of trainer.py:619-635."""
        return {
            'epoch': info['epoch'],
            'model_state_dict': transformer.state_dict(),
            "decoder_state_dict": decoder.state_dict() if decoder is not None else None,
            'optimizer_state_dict': transformer_optimizer.state_dict(),
            "optim_name": self.optim_name,
            'loss': info['total'],
            'completed_step': info['completed_step'],
            "criterion": {"alpha": self.alpha, "beta": self.beta, "gamma": self.gamma, "delta": self.delta,
                          "content_layers_idx": self.content_layers_idx, "style_layers_idx": self.style_layers_idx},
        }

    def save(self, transformer, transformer_optimizer, decoder=None, info=None, result=None):
        """trainer.py:607-655: TotalL_.../transformer{N}.pth + config text; rank 0 only under DP."""
        if self.rank != 0:
            return None
        dir_name = f"TotalL_{info['total']}_Content_{info['content']}_Style_{info['style']}"
        save_path_dir = os.path.join(self.output_dir, dir_name)
        print(f" Saving in {save_path_dir}")
        if not os.path.exists(save_path_dir):
            os.makedirs(save_path_dir, exist_ok=False)
        else:
            warnings.warn("File exist!, might be a good idea to change seed")
            raise FileExistsError
        n_examples = len(self.dataloaders['train'].dataset) if hasattr(self.dataloaders['train'], 'dataset') \
            else len(self.dataloaders['train']) * self.per_device_batch_size
        model_path = os.path.join(save_path_dir, f"transformer{n_examples}" + ".pth")
        torch.save(self.checkpoint_dict(transformer, transformer_optimizer, decoder, info), model_path)
        try:
            stamp = '_'.join(str(datetime.datetime.now().strftime('%Y-%m-%d %H-%M-%S')).split())
            with open(os.path.join(save_path_dir, f"config_E{info['epoch']}_{stamp}"), "w") as f:
                f.write(f"\n          {PRJ_NAME}\n")
                f.write(f"\n   Epoch: {info['epoch']}")
                f.write(f"\n   Completed step: {info['completed_step']}")
                f.write(f"\n   Loss: {info['total']}")
                for key, value in vars(self.config).items():
                    f.write(f"\n {key}: {value} ")
        except IOError:
            warnings.warn(f"Can't save config for this run {info['epoch']}")
        if result is not None:
            for idx, plot in enumerate(result):
                plot.save(os.path.join(save_path_dir, f"result_plot{idx}.png"))
        return model_path

    def resume(self, transformer, decoder, transformer_optimizer):
        """Resume rules of trainer.py:304-378.  Returns (optimizer, init_epoch, loss_list,
        completed_step, last_session_epoch)."""
        init_epoch, loss_list, completed_step, last_session_epoch = 0, [], 0, -1
        if self.resume_from_checkpoint is None:
            return transformer_optimizer, init_epoch, loss_list, completed_step, last_session_epoch
        try:
            checkpoint = torch.load(self.resume_from_checkpoint, map_location=self.device, weights_only=True)
        except Exception:
            raise RuntimeError("Unable to load from checkpoint, invalid path!")
        try:
            transformer.load_state_dict(checkpoint['model_state_dict'])
        except Exception:
            raise RuntimeError("Unable to load weight to the model, wrong model structure compare to checkpoint!")
        try:
            if checkpoint['decoder_state_dict'] is not None:
                decoder.load_state_dict(checkpoint['decoder_state_dict'])
        except Exception:
            raise RuntimeError("Unable to load weight to the decoder model, wrong model structure compare to checkpoint!")
        try:
            ck_name = checkpoint['optim_name']['optim_name']
            if self.optim_name['optim_name'] != ck_name:
                params = list(transformer.parameters()) + (list(decoder.parameters()) if self.do_decoder_train else [])
                transformer_optimizer = self.get_optimizer(ck_name, params, self.learning_rate, checkpoint['optim_name'])
                warnings.warn(f"Set optim {self.optim_name['optim_name']} different from checkpoint optim {ck_name},"
                              f" switching to {ck_name}")
                self.optim_name['optim_name'] = ck_name
            transformer_optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
            self._retarget_optimizer(transformer_optimizer)
        except Exception:
            raise RuntimeError("Unable to load optimizer state!")
        try:
            criterion = checkpoint['criterion']
            if all(criterion[k] == getattr(self, k) for k in
                   ('alpha', 'beta', 'gamma', 'delta', 'content_layers_idx', 'style_layers_idx')):
                loss_list.append(checkpoint['loss'])
            else:
                warnings.warn("Last session loss discarded due to changed loss criteria")
        except KeyError:
            warnings.warn("checkpoint missing criterion info!")
        try:
            completed_step += checkpoint['completed_step']
            last_session_epoch = checkpoint['epoch']
            if last_session_epoch + 1 < self.num_train_epochs:
                init_epoch = last_session_epoch + 1
            else:
                raise RuntimeError("Num train epoch can't be smaller than last session epoch resume from checkpoint!")
        except KeyError:
            warnings.warn("Checkpoint missing train process info!")
        return transformer_optimizer, init_epoch, loss_list, completed_step, last_session_epoch

    def _retarget_optimizer(self, optimizer) -> None:
        """``load_state_dict`` overwrites ``param_groups`` with the checkpoint's: a checkpoint written by an eager
        run has ``capturable=False``, a float lr and host-side step counters, one written by a graph run the
        opposite.  Put the loaded optimizer into the form THIS run needs (graph replay: capturable, lr and step
        counters on the device so the scheduler's ``lr.fill_`` reaches the captured kernels; eager: plain floats)."""
        graph = self.graph_capable()
        for group in optimizer.param_groups:
            lr = group['lr']
            if graph:
                group['capturable'] = True
                if not isinstance(lr, torch.Tensor):
                    group['lr'] = torch.tensor(float(lr), dtype=torch.float32, device=self.device)
                elif lr.device != self.device:
                    group['lr'] = lr.to(self.device)
                if 'initial_lr' in group and isinstance(group['initial_lr'], torch.Tensor):
                    group['initial_lr'] = float(group['initial_lr'])
            else:
                if 'capturable' in group:
                    group['capturable'] = False
                if isinstance(lr, torch.Tensor):
                    group['lr'] = float(lr)
                if isinstance(group.get('initial_lr'), torch.Tensor):
                    group['initial_lr'] = float(group['initial_lr'])
        for p, st in optimizer.state.items():
            step = st.get('step')
            if isinstance(step, torch.Tensor):
                want = p.device if graph else torch.device('cpu')
                if step.device != want:
                    st['step'] = step.to(want)

    # ---- training -------------------------------------------------------------------------
    def make_step(self, encoder, decoder, transformer, loss_vgg) -> StyleTransferStep:
        return StyleTransferStep(encoder, decoder, transformer, loss_vgg, self.content_layers_idx,
                                 self.style_layers_idx, self.alpha, self.beta, self.gamma, self.delta,
                                 self.use_VincentStyleLoss)

    def trainable_parameters(self, transformer, decoder):
        params = list(transformer.parameters())
        if self.do_decoder_train:
            params += list(decoder.parameters())          # trainer.py:286-291
        else:
            for p in decoder.parameters():
                p.requires_grad_(False)                   # never read by an optimizer (trainer.py:292-297)
        return params

    # The step in three phases with the two data-parallel exchanges between them (GraphedStep replays each phase
    # from its own HIP graph; eager steps call them back to back).
    def step_forward(self, step: StyleTransferStep, bucket: GradBucket, content_imgs, style_imgs, prefetch=None):
        """zero_grad (one memset of the flat bucket) + the forward up to the per-rank Hellinger sums.
        ``prefetch``: the NEXT batch (content, style), see optimize_step."""
        bucket.zero()
        # Weight gradients are written straight into the bucket -- an OVERWRITE, valid for exactly one decoder backward
        # per zeroed bucket.  The flag is therefore scoped to this step: set here, cleared by step_backward, so any
        # other backward through this decoder (a caller accumulating over several backwards, the decoder used twice
        # in one graph) goes through autograd's accumulating path.
        step.decoder._packed.direct_grads = True
        front = self._take_front(content_imgs, style_imgs)
        hook = None
        if prefetch is not None and hasattr(step.style_loss, "prefetch_stream"):
            hook = lambda: self._prefetch_front(step, *prefetch)        # noqa: E731
        return step.forward_partials(content_imgs, style_imgs, front, hook)

    # ---- the next batch's weight-independent forward, computed under this batch's eigen-decompositions -----------
    def _prefetch_front(self, step: StyleTransferStep, content_imgs, style_imgs):
        """Called by VincentStyleLoss.forward_multi once this step's eigen jobs are on their lanes: StyleTransferStep.
        front of the next batch on the free side stream (it waits for everything issued on the current stream so far,
        i.e. for the batch's tensors, and for nothing the lanes do)."""
        cur = torch.cuda.current_stream()
        pf = step.style_loss.prefetch_stream()
        pf.wait_stream(cur)
        with torch.cuda.stream(pf):
            fr = step.front(content_imgs, style_imgs, inline_targets=True)
        self._front_next = (content_imgs, style_imgs, fr, pf)

    def _take_front(self, content_imgs, style_imgs):
        """The prefetched front of exactly these batch tensors, made safe for the current stream; else None."""
        nxt, self._front_next = getattr(self, "_front_next", None), None
        if nxt is None or nxt[0] is not content_imgs or nxt[1] is not style_imgs:
            return None
        _, _, fr, pf = nxt
        cur = torch.cuda.current_stream()
        cur.wait_stream(pf)

        def claim(v):                                   # allocated on the prefetch stream, used (and freed) here
            if isinstance(v, torch.Tensor):
                v.record_stream(cur)
            elif isinstance(v, (list, tuple)):
                for w in v:
                    claim(w)
        for v in fr.values():
            claim(v)
        return fr

    def step_backward(self, step: StyleTransferStep, partials, q_sum, global_batch: int):
        """Rest of compute_loss from the batch-wide q, then backward into the bucket (trainer.py:472)."""
        q = _InjectReducedQ.apply(partials[3], q_sum, self.world_size) if self.dp_active else partials[3]
        losses = step.combine(partials, q, global_batch)
        try:
            losses[4].backward()
        finally:
            step.decoder._packed.direct_grads = False
        return losses

    def step_update(self, step: StyleTransferStep, optimizer, bucket: GradBucket):
        """[1 / world_size] / [AdaIN grad clamp, AFTER the reduction] / optimizer.step (trainer.py:392-394, 473)."""
        if self.dp_active:
            bucket.flat.div_(self.world_size)
        if self.gradient_threshold is not None:
            for p in step.transformer.parameters():
                if p.grad is not None:
                    p.grad.clamp_(-self.gradient_threshold, self.gradient_threshold)
        # an eps parameter in a clamped branch of adjust_sigmoid gets no gradient in the reference (transformer.py:
        # 32-35 returns a Python float) and is therefore never moved by the optimizer; keep it bit-identical here too
        # (weight decay or left-over momentum would otherwise move it on a zero gradient) -- device-side, no host read
        eps_params = (step.transformer.content_eps, step.transformer.style_eps)
        masks = step.transformer.clamped_masks()
        before = [p.detach().clone() for p in eps_params]
        optimizer.step()
        with torch.no_grad():
            for p, old, m in zip(eps_params, before, masks):
                p.copy_(torch.where(m, old, p))

    def optimize_step(self, step: StyleTransferStep, optimizer, bucket: GradBucket, content_imgs, style_imgs,
                      prefetch=None):
        """zero_grad / backward / [grad all-reduce] / [AdaIN grad clamp] / optimizer.step (trainer.py:471-473, 392-394).
        ``prefetch`` = the NEXT batch's (content, style) device tensors, or None.  With VincentStyleLoss the part of
        that batch's forward which touches no trained weight (Normalize, both loss-VGG passes over the originals, the
        frozen encoder, cov_stl^(1/2)) is issued on a side stream while THIS batch's eigen-decompositions run -- a
        13 ms chain of tiny launches that leaves the chip idle -- and is picked up when the next call gets exactly
        those tensors.  Same values as computing it in the next step; other style losses ignore the argument."""
        partials = self.step_forward(step, bucket, content_imgs, style_imgs, prefetch)
        q_sum = partials[3]
        if self.dp_active:
            dist = _dist()
            q_sum = partials[3].detach().clone()
            dist.all_reduce(q_sum, op=dist.ReduceOp.SUM)
        losses = self.step_backward(step, partials, q_sum, content_imgs.shape[0] * self.world_size)
        if self.dp_active:
            bucket.all_reduce_sum()
        self.step_update(step, optimizer, bucket)
        return losses

    def train(self):
        from . import precision
        precision.set_compute_dtype(self.compute_dtype)
        encoder, decoder, transformer, loss_vgg = self.build_model()
        params = self.trainable_parameters(transformer, decoder)
        # graph replay: the learning rate lives in a device tensor so that the scheduler's updates (lr.fill_)
        # reach the captured optimizer kernels
        lr = torch.tensor(float(self.learning_rate), device=self.device) if self.graph_capable() else self.learning_rate
        transformer_optimizer = self.get_optimizer(self.optim_name['optim_name'], params, lr,
                                                   self.graph_optim_kwargs())
        transformer_optimizer, init_epoch, loss_list, completed_step, last_session_epoch = \
            self.resume(transformer, decoder, transformer_optimizer)
        train_loader = self.dataloaders['train']
        steps_per_epoch = len(train_loader)
        scheduler = lr_scheduler.CosineAnnealingLR(
            transformer_optimizer, T_max=steps_per_epoch * (self.num_train_epochs - init_epoch),
            last_epoch=last_session_epoch if self.resume_from_checkpoint else -1)
        scheduler_steps = math.ceil(steps_per_epoch * self.step_frequency)       # trainer.py:387
        scheduler_count = 0
        step_engine = self.make_step(encoder, decoder, transformer, loss_vgg)
        bucket = GradBucket(params)
        run_step = GraphedStep(self, step_engine, transformer_optimizer, bucket) if self.graph_capable() else \
            (lambda c, s: self.optimize_step(step_engine, transformer_optimizer, bucket, c, s))
        history = []
        if self.rank == 0:       # what the reference announces before its loop (trainer.py:396-413), one line per fact
            n_examples = len(train_loader.dataset) if hasattr(train_loader, "dataset") else "?"
            facts = [("epochs", self.num_train_epochs), ("first epoch of this run", init_epoch),
                     ("style loss", "VincentStyleLoss" if self.use_VincentStyleLoss else "AdaINStyleLoss"),
                     ("optimizer", self.optim_name), ("gradient threshold", self.gradient_threshold),
                     ("grayscale content", self.grayscale_content_transform), ("eps limit", self.eps),
                     ("decoder trained", self.do_decoder_train), ("WCT decoder weights", self.use_pretrained_WCTDECODER),
                     ("batch per process", self.per_device_batch_size), ("processes", self.world_size),
                     ("batches per epoch", steps_per_epoch), ("examples", n_examples),
                     ("optimizer steps in total", steps_per_epoch * (self.num_train_epochs - init_epoch)),
                     ("batches between LR steps", scheduler_steps), ("content layers", self.content_layers_idx),
                     ("style layers", self.style_layers_idx), ("device", self.device),
                     ("conv compute", self.compute_dtype), ("HIP-graph replay", self.graph_capable())]
            print("\n --- training set-up ---\n" + "\n".join(f"   {k}: {v}" for k, v in facts) + "\n")
        epochs = range(init_epoch, self.num_train_epochs)
        if self.rank == 0:
            try:        # the reference's epoch progress bar (trainer.py:416-417)
                from tqdm import tqdm
                epochs = tqdm(epochs, desc="Training progress", colour='green', position=0, leave=True)
            except ImportError:
                pass
        for epoch in epochs:
            sums = torch.zeros(5, dtype=torch.float64, device=self.device)
            step_log = torch.zeros((steps_per_epoch, 5), dtype=torch.float32, device=self.device) if self.log_steps else None
            transformer.train()
            if self.do_decoder_train:
                decoder.train()
            # one batch of lookahead: with VincentStyleLoss the next batch's weight-independent forward is issued under
            # this batch's eigen-decompositions (optimize_step(prefetch=...)); HIP-graph replay has no such stall
            lookahead = self.use_VincentStyleLoss and not self.graph_capable()

            def on_device(batch):
                c_ = batch['content_image'].to(self.device, non_blocking=True).contiguous()
                s_ = batch['style_image'].to(self.device, non_blocking=True).contiguous()
                if self.gpu_augment is not None:
                    c_, s_ = self.gpu_augment[0](c_), self.gpu_augment[1](s_)
                return c_, s_
            batches = iter(train_loader)
            nxt = next(batches, None)
            nxt = on_device(nxt) if nxt is not None else None
            step = -1
            while nxt is not None:
                step += 1
                content_imgs, style_imgs = nxt
                nxt = next(batches, None)
                nxt = on_device(nxt) if nxt is not None else None
                if lookahead:
                    losses = self.optimize_step(step_engine, transformer_optimizer, bucket, content_imgs, style_imgs,
                                                prefetch=nxt)
                else:
                    losses = run_step(content_imgs, style_imgs)
                completed_step += step                                          # trainer.py:475 (sic: index, not 1)
                row = torch.stack([l.detach().reshape(()) for l in losses])
                sums += row.double()
                if step_log is not None and step < steps_per_epoch:
                    step_log[step].copy_(row)
                scheduler_count += 1
                if scheduler_count >= scheduler_steps:                          # trainer.py:498-502
                    scheduler.step()
                    scheduler_count = 0
            if self.use_VincentStyleLoss:
                from . import eigh
                eigh.check()                                                     # solver `info` of the whole epoch
            avg = (sums / steps_per_epoch).tolist()                              # one device->host read per epoch
            avg_content, avg_style, avg_var, avg_hist, avg_total = avg
            loss_list.append(avg_total)
            history.append({"epoch": epoch, "content": avg_content, "style": avg_style, "variation": avg_var,
                            "hist": avg_hist, "total": avg_total})
            if step_log is not None:
                history[-1]["steps"] = step_log.tolist()       # [step][content, style, variation, hist, total]
                if self.rank == 0:
                    for i, (c_, s_, v_, h_, t_) in enumerate(history[-1]["steps"]):
                        print(f"     step {i}: content {c_:.6g} style {s_:.6g} variation {v_:.6g} hist {h_:.6g} "
                              f"total {t_:.6g}")
            if self.rank == 0:       # the per-epoch report of trainer.py:514-525: raw and weighted terms, eps, best so far
                epss = transformer.get_current_eps()
                rows = [("content", avg_content, self.alpha), ("style", avg_style, self.beta),
                        ("variation", avg_var, self.gamma), ("histogram", avg_hist, self.delta)]
                print(f"\n --- epoch {epoch + 1} of {self.num_train_epochs} ---\n"
                      f"   total loss {avg_total}   (best of this session: {min(loss_list)})\n" +
                      "".join(f"   {name:9s} {val}   weighted {val * wgt}\n" for name, val, wgt in rows) +
                      f"   eps in use: style {epss[0]} (parameter {float(transformer.style_eps.detach())}), "
                      f"content {epss[1]} (parameter {float(transformer.content_eps.detach())})")
            info = {"total": avg_total, "content": avg_content, "style": avg_style, "epoch": epoch,
                    "completed_step": completed_step}
            plots = None
            if self.do_eval_per_epoch and avg_total == min(loss_list) and self.save_best and self.rank == 0:
                plots = self.evaluate_plots(encoder, decoder, transformer)       # trainer.py:530-559
            if (avg_total == min(loss_list) and self.save_best) or not self.save_best:   # trainer.py:580-595
                self.save(transformer, transformer_optimizer, decoder, info=info, result=plots)
        return history

    # ---- inference composition --------------------------------------------------------------
    @staticmethod
    @torch.no_grad()
    def plot_comparison(encoder, decoder, transformer, content_img_url: list, style_img_url: list, device='cpu',
                        content_transformation=None, style_transformation=None, sleep: int = 5,
                        plot: bool = True, alpha: float = 1.0):
        """Trainer.plot_comparison (trainer.py:672-747): like the reference, only the LAST (content, style)
        pair of the two lists is used; returns (pyplot module holding the 10x4 in figure, normalised decoded
        image [1,3,H,W]).  The blend of :702 is fused into the AdaIN kernel."""
        from PIL import Image
        import numpy as np
        from . import transforms as T
        content_image = style_image = None
        for content_url, style_url in zip(content_img_url, style_img_url):
            content_image = Image.open(content_url).convert('RGB')
            style_image = Image.open(style_url).convert('RGB')
        content_image_tensor = content_transformation(content_image).to(device)
        content_image_tensor_org = T.Compose([T.Resize(256), T.ToTensor()])(content_image)
        style_image_tensor = style_transformation(style_image).to(device)
        transformer.eval()
        decoder.eval()
        decode_img = Trainer.stylize(encoder, decoder, transformer, content_image_tensor.unsqueeze(0).contiguous(),
                                     style_image_tensor.unsqueeze(0).contiguous(), alpha=alpha, normalize=True)
        import matplotlib
        if not plot:
            matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        imgs = [content_image_tensor_org.squeeze().permute(1, 2, 0).detach().cpu().numpy(),
                style_image_tensor.squeeze().permute(1, 2, 0).detach().cpu().numpy(),
                decode_img.squeeze().permute(1, 2, 0).detach().cpu().numpy()]
        fig, axes = plt.subplots(nrows=1, ncols=3, figsize=(10, 4))
        for ax, im, title in zip(axes, imgs, ('Content', 'Style', 'Stylized content')):
            ax.imshow(np.interp(im, (im.min(), im.max()), (0, 1)))
            ax.set_title(title)
            ax.axis('off')
        plt.tight_layout()
        if plot:
            plt.show(block=False)
            plt.pause(sleep)
        return plt, decode_img

    def evaluate_plots(self, encoder, decoder, transformer):
        """The per-epoch eval grid of trainer.py:530-559: one 1000x400 RGBA comparison per eval batch."""
        from . import transforms as T
        plots = []
        for batch in self.dataloaders.get('eval', []):
            plot, _ = self.plot_comparison(
                encoder, decoder, transformer, batch['content_image'], batch['style_image'],
                content_transformation=T.Compose([T.Resize(256), T.ToTensor(),
                                                  T.RGBToGrayscaleStacked(enable=self.grayscale_content_transform)]),
                style_transformation=T.Compose([T.Resize(256), T.ToTensor()]),
                device=self.device, plot=self.plot_per_epoch)
            try:
                from .inference import figure_to_image
                plots.append(figure_to_image(plot))
            except Exception:
                warnings.warn("Byte IO fail!")
        return plots

    @staticmethod
    @torch.no_grad()
    def stylize(encoder, decoder, transformer, content: torch.Tensor, style: torch.Tensor, alpha: float = 1.0,
                normalize: bool = True):
        """The tensor part of plot_comparison (trainer.py:695-708) on batched inputs."""
        from .pipeline import stylize
        out = stylize(encoder, decoder, transformer, content, style, alpha)
        return Normalize(IMAGENET_MEAN, IMAGENET_STD)(out) if normalize else out
