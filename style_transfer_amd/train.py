"""``src/train.py`` of the reference (argument surface :13-101, main :104-173) on the HIP hot path.

    python -m style_transfer_amd.train --output_dir OUT --content_datapath DIR.. --style_datapath DIR.. [...]
    python -m torch.distributed.run --nproc-per-node N -m style_transfer_amd.train ...      (data parallel)

Same flags and defaults as the reference; additions: ``--encoder_weights``, ``--decoder_weights``,
``--vgg_weights`` (the blobs the reference hard-codes or downloads), ``--dtype f32|bf16|bf16x3``, ``--hip_graph``.
Under ``torch.distributed.run`` one process drives each GPU (RCCL), the sampler is sharded by rank and
rank 0 writes the checkpoints.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

from . import transforms as T
from .dataloader import STDataloader
from .trainer import Trainer
from .utils import ParseKwargsOptim, clear_cuda_cache, set_seed


# The reference's flag schema (src/train.py:13-87): names, types and defaults are the drop-in contract; the help text is ours.
_FLAGS = [
    # data
    ('--output_dir', dict(type=str, help="directory that receives TotalL_*/transformer*.pth checkpoints")),
    ('--content_datapath', dict(nargs='+', type=str, help="one or more directories of content images")),
    ('--style_datapath', dict(nargs='+', type=str, help="one or more directories of style images")),
    ('--eval_contentpath', dict(type=str, default="./src/data/eval_dir/content", help="content images of the eval grid")),
    ('--eval_stylepath', dict(type=str, default="./src/data/eval_dir/style", help="style images of the eval grid")),
    ('--batch_size', dict(type=int, default=2, help="training batch per process (per GPU under data parallel)")),
    ('--eval_batch_size', dict(type=int, default=1, help="eval loader batch (paths, not tensors)")),
    ('--max_content_train_samples', dict(type=int, default=10000, help="cap on content images used")),
    ('--max_style_train_samples', dict(type=int, default=10000, help="cap on style images used")),
    ('--num_worker', dict(type=int, default=2, help="DataLoader worker processes")),
    ('--seed', dict(type=int, default=42, help="seed for python / numpy / torch and the sampler")),
    ('--crop_width', dict(type=int, default=256, help="random-crop width after Resize(300)")),
    ('--crop_height', dict(type=int, default=256, help="random-crop height after Resize(300)")),
    ('--recursive_paths_load', dict(action='store_true', help="also collect images from sub-directories")),
    # schedule / run control
    ('--num_train_epochs', dict(type=int, default=10, help="epochs to run (counted from 0, also when resuming)")),
    ('--step_frequency', dict(type=float, default=0.5, help="cosine LR step every this fraction of an epoch")),
    ('--with_tracking', dict(action='store_true', help="wandb tracking (not part of this build: raises)")),
    ('--log_weights_cpkt', dict(action='store_true', help="upload checkpoints to the tracker")),
    ('--resume_from_checkpoint', dict(type=str, default=None, help="transformer*.pth to continue from")),
    ('--plot_per_epoch', dict(action='store_true', help="show the comparison figure after each epoch")),
    ('--do_eval_per_epoch', dict(action='store_true', help="render the eval grid when an epoch is the best so far")),
    ('--do_decoder_train', dict(action='store_true', help="train the decoder too (otherwise only AdaIN's eps)")),
    ('--use_pretrained_WCTDECODER', dict(action='store_true', help="start the decoder from feature_invertor_conv4_1.pth")),
    ('--grayscale_content_transform', dict(action='store_true', help="feed content images as stacked grayscale")),
    # optimiser / losses
    ('--vgg_model_type', dict(type=str, default='19', help="loss network family (only '19' is ever built)")),
    ('--optim_name', dict(nargs='+', default={'optim_name': 'adam'}, action=ParseKwargsOptim,
                          help="optimizer name followed by key=value kwargs, e.g. adamax betas=0.9,0.99")),
    ('--learning_rate', dict(type=float, default=1e-4, help="initial learning rate of the cosine schedule")),
    ('--gradient_threshold', dict(type=float, default=None, help="clamp AdaIN eps gradients to +-this value")),
    ('--alpha', dict(type=float, default=1, help="content-loss weight")),
    ('--beta', dict(type=float, default=10, help="style-loss weight")),
    ('--gamma', dict(type=float, default=1, help="total-variation weight")),
    ('--delta', dict(type=float, default=10, help="RGB-uv histogram-loss weight")),
    ('--eps', dict(type=float, default=1e-5, help="upper limit of AdaIN's trainable eps")),
    ('--content_layers_idx', dict(nargs='+', type=int, default=[21], help="vgg19.features indices of the content loss")),
    ('--style_layers_idx', dict(nargs='+', type=int, default=[2, 7, 14, 19, 25], help="vgg19.features indices of the style loss")),
    ('--use_VincentStyleLoss', dict(action='store_true', help="covariance / L2-Wasserstein style loss instead of mean-std")),
    # additions (not in the reference)
    ('--encoder_weights', dict(type=str, default=None, help="vgg_normalised_conv4_1.pth")),
    ('--decoder_weights', dict(type=str, default=None, help="feature_invertor_conv4_1.pth")),
    ('--vgg_weights', dict(type=str, default=None, help="torchvision vgg19 state dict (vgg19-dcbb9e9d.pth)")),
    ('--dtype', dict(choices=["f32", "bf16", "bf16x3"], default="f32", help="conv compute precision")),
    ('--hip_graph', dict(action='store_true', help="replay the training step from HIP graphs (also under data parallel)")),
    ('--gpu_augment', dict(action='store_true',
                           help="run the post-ToTensor augmentations (rotation, noise, sharpness, colour jitter, flip) "
                                "on the GPU per batch (gpu_augment.py) instead of per image in the DataLoader workers")),
    ('--log_steps', dict(action='store_true',
                         help="keep the five per-batch loss scalars (trainer.py:475-480) and print them at the end of "
                              "each epoch; they stay on the device during the epoch, no per-step host read")),
]


def parse_args(args):
    parser = argparse.ArgumentParser(description="AdaIN style-transfer training on the MI355X HIP hot path")
    for flag, spec in _FLAGS:
        parser.add_argument(flag, **spec)
    args = parser.parse_args(args)
    # what the reference refuses to start with (train.py:89-99), checked up front
    problems = []
    if not os.path.isdir(args.output_dir or ""):
        problems.append("Invalid output dir path!")
    if not all(os.path.isdir(p) for p in args.content_datapath or [""]):
        problems.append("Invalid content dir path!")
    if not all(os.path.isdir(p) for p in args.style_datapath or [""]):
        problems.append("Invalid style dir path!")
    if args.resume_from_checkpoint is not None and not os.path.isfile(args.resume_from_checkpoint):
        problems.append("Invalid cpkt path!")
    if min(args.max_content_train_samples, args.max_style_train_samples) <= 0:
        problems.append("Number of examples must be higher than 0")
    assert not problems, " ".join(problems)
    return args


def build_transforms(args):
    """train.py:116-135.  (crop_width, crop_height) is passed as torchvision's (h, w) size, as there.
    With ``--gpu_augment`` the workers stop after ToTensor; build_gpu_augment() supplies the rest."""
    crop = T.RandomCrop((args.crop_width, args.crop_height), pad_if_needed=True, padding=1)
    if getattr(args, "gpu_augment", False):
        return T.Compose([T.Resize(300), crop, T.ToTensor()]), T.Compose([T.Resize(300), crop, T.ToTensor()])
    content = T.Compose([
        T.Resize(300), crop, T.ToTensor(), T.RGBToGrayscaleStacked(enable=args.grayscale_content_transform),
        T.RandomRotation(degrees=90), T.RandomAdjustSharpness(sharpness_factor=1.5, p=0.5),
        T.ColorJitter(brightness=0.2, contrast=0.4, saturation=0.3, hue=0.1), T.RandomHorizontalFlip(p=0.5)])
    style = T.Compose([
        T.Resize(300), crop, T.ToTensor(), T.RandomRotation(degrees=90),
        T.AddGaussianNoise(mean=0.5, sigma_range=(0., 0.08), p=0.5),
        T.RandomAdjustSharpness(sharpness_factor=1.5, p=0.5),
        T.ColorJitter(brightness=0.2, contrast=0.25, saturation=0.3, hue=0.1), T.RandomHorizontalFlip(p=0.5)])
    return content, style


def build_gpu_augment(args, rank: int = 0):
    """The tensor half of train.py:116-135 as two GpuAugment objects (content, style), or None."""
    if not getattr(args, "gpu_augment", False):
        return None
    from .gpu_augment import GpuAugment
    return (GpuAugment.for_content(grayscale=args.grayscale_content_transform, seed=2 * (args.seed + 7919 * rank)),
            GpuAugment.for_style(seed=2 * (args.seed + 7919 * rank) + 1))


def main(args):
    args = parse_args(args)
    rank, world = 0, int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:                                    # one process per GPU, RCCL over xGMI
        from .parallel import init_process_group
        dist, _ = init_process_group(rehearsal=bool(os.environ.get("ST_DP_REHEARSAL")))
        rank = dist.get_rank()
    clear_cuda_cache()
    set_seed(args.seed)

    transform_content, transform_style = build_transforms(args)
    dataloaders = STDataloader(
        content_datapath=args.content_datapath, style_datapath=args.style_datapath,
        eval_contentpath=args.eval_contentpath, eval_stylepath=args.eval_stylepath,
        batch_size=args.batch_size, eval_batch_size=args.eval_batch_size,
        transform_content=transform_content, transform_style=transform_style,
        max_style_train_samples=args.max_style_train_samples, max_content_train_samples=args.max_content_train_samples,
        recursive_paths_load=args.recursive_paths_load, num_worker=args.num_worker, rank=rank, world_size=world)

    trainer = Trainer(
        dataloaders=dataloaders, output_dir=args.output_dir, resume_from_checkpoint=args.resume_from_checkpoint,
        with_tracking=args.with_tracking, num_train_epochs=args.num_train_epochs,
        per_device_batch_size=dataloaders.batch_size, do_eval_per_epoch=args.do_eval_per_epoch,
        plot_per_epoch=args.plot_per_epoch, learning_rate=args.learning_rate, vgg_model_type=args.vgg_model_type,
        seed=args.seed, alpha=args.alpha, beta=args.beta, gamma=args.gamma, delta=args.delta, eps=args.eps,
        content_layers_idx=args.content_layers_idx, style_layers_idx=args.style_layers_idx,
        log_weights_cpkt=args.log_weights_cpkt, step_frequency=args.step_frequency, optim_name=args.optim_name,
        gradient_threshold=args.gradient_threshold, do_decoder_train=args.do_decoder_train,
        use_pretrained_WCTDECODER=args.use_pretrained_WCTDECODER,
        grayscale_content_transform=args.grayscale_content_transform, use_VincentStyleLoss=args.use_VincentStyleLoss,
        config=args, encoder_weights=args.encoder_weights, decoder_weights=args.decoder_weights,
        vgg_weights=args.vgg_weights, compute_dtype=args.dtype, use_hip_graph=args.hip_graph,
        log_steps=args.log_steps, gpu_augment=build_gpu_augment(args, rank))
    history = trainer.train()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    main(sys.argv[1:])
