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


def parse_args(args):
    parser = argparse.ArgumentParser()
    # Data
    parser.add_argument('--output_dir', type=str, help="The output directory to save")
    parser.add_argument('--content_datapath', nargs='+', type=str,
                        help="The path to content images dir (can be a list of paths)")
    parser.add_argument('--style_datapath', nargs='+', type=str,
                        help="The path to style images dir (can be a list of paths)")
    parser.add_argument('--eval_contentpath', type=str, default="./src/data/eval_dir/content", help="The path to eval dir")
    parser.add_argument('--eval_stylepath', type=str, default="./src/data/eval_dir/style", help="The path to style dir")
    parser.add_argument('--batch_size', type=int, default=2, help="Batch size for the dataloader")
    parser.add_argument('--eval_batch_size', type=int, default=1, help="Eval batch size for the dataloader")
    parser.add_argument('--max_content_train_samples', type=int, default=10000, help="Number of content training samples")
    parser.add_argument('--max_style_train_samples', type=int, default=10000, help="Number of style training samples")
    parser.add_argument('--num_worker', type=int, default=2, help="Number of worker for dataloader")
    parser.add_argument('--seed', type=int, default=42, help="A seed for reproducible training.")
    parser.add_argument('--crop_width', type=int, default=256, help="Width of the image randomly center crop")
    parser.add_argument('--crop_height', type=int, default=256, help="Width of the image randomly center crop")
    parser.add_argument('--recursive_paths_load', action='store_true',
                        help="Whether to enable recursive load to load all instance of image files"
                             "found in the content and style datapath.")
    # Training
    parser.add_argument('--num_train_epochs', type=int, default=10, help="number training epochs")
    parser.add_argument('--step_frequency', type=float, default=0.5,
                        help="How often should you update the lr (Should be a fraction of an epoch)")
    parser.add_argument('--with_tracking', action='store_true', help="Whether to enable experiment trackers for logging.")
    parser.add_argument('--log_weights_cpkt', action='store_true', help="Whether to log weights to tracker")
    parser.add_argument('--resume_from_checkpoint', type=str, default=None,
                        help="If the training should continue from a checkpoint folder. (can be bool or string)")
    parser.add_argument('--plot_per_epoch', action='store_true', help="Whether to plot comparison per epoch.")
    parser.add_argument('--do_eval_per_epoch', action='store_true', help="Whether to run evaluate per epoch.")
    parser.add_argument('--do_decoder_train', action='store_true',
                        help="Whether to enable decoder training to expand model capacity.")
    parser.add_argument('--use_pretrained_WCTDECODER', action='store_true',
                        help="Whether to load pretrained WCT decoder that were trained on image recovery.")
    parser.add_argument('--grayscale_content_transform', action='store_true',
                        help="Whether to enable grayscale content convert.")
    # Optimizer
    parser.add_argument('--vgg_model_type', type=str, default='19',
                        help="Which models of the vgg to use as a feature extractor")
    parser.add_argument('--optim_name', nargs='+', default={'optim_name': 'adam'}, action=ParseKwargsOptim,
                        help="Which optimizer to use, support for Kwargrs, example: '--optim_name adamax betas=0.9,0.99' ")
    parser.add_argument('--learning_rate', type=float, default=1e-4,
                        help="Initial learning rate (after the potential warmup period) to use.")
    parser.add_argument('--gradient_threshold', type=float, default=None,
                        help="Gradient threshold for clipping (use for exploding gradient) (recommended range 1-10)")
    parser.add_argument('--alpha', type=float, default=1, help="weight for the content loss.")
    parser.add_argument('--beta', type=float, default=10, help="weight for style loss.")
    parser.add_argument('--gamma', type=float, default=1, help="weight for variance loss.")
    parser.add_argument('--delta', type=float, default=10, help="weight for histogram loss.")
    parser.add_argument('--eps', type=float, default=1e-5, help="Maximum eps value for AdaIN for both content and style.")
    parser.add_argument('--content_layers_idx', nargs='+', type=int, default=[21], help="Vgg layers to compute content loss")
    parser.add_argument('--style_layers_idx', nargs='+', type=int, default=[2, 7, 14, 19, 25],
                        help="Vgg layers to compute style loss")
    parser.add_argument('--use_VincentStyleLoss', action='store_true',
                        help="Whether to enable the default style loss implementation using Gram matrices")
    # additions (not in the reference)
    parser.add_argument('--encoder_weights', type=str, default=None, help="vgg_normalised_conv4_1.pth")
    parser.add_argument('--decoder_weights', type=str, default=None, help="feature_invertor_conv4_1.pth")
    parser.add_argument('--vgg_weights', type=str, default=None, help="torchvision vgg19 state dict (vgg19-dcbb9e9d.pth)")
    parser.add_argument('--dtype', choices=["f32", "bf16", "bf16x3"], default="f32", help="conv compute precision")
    parser.add_argument('--hip_graph', action='store_true',
                        help="capture the training step in a HIP graph and replay it (single process)")
    args = parser.parse_args(args)

    # Sanity check (train.py:89-99)
    assert os.path.isdir(args.output_dir), "Invalid output dir path!"
    for path in args.content_datapath:
        assert os.path.isdir(path), "Invalid content dir path!"
    for path in args.style_datapath:
        assert os.path.isdir(path), "Invalid style dir path!"
    assert os.path.isfile(args.resume_from_checkpoint) if args.resume_from_checkpoint is not None else True, \
        "Invalid cpkt path!"
    assert args.max_content_train_samples > 0 and args.max_style_train_samples > 0, \
        "Number of examples must be higher than 0"
    return args


def build_transforms(args):
    """train.py:116-135.  (crop_width, crop_height) is passed as torchvision's (h, w) size, as there."""
    crop = T.RandomCrop((args.crop_width, args.crop_height), pad_if_needed=True, padding=1)
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


def main(args):
    args = parse_args(args)
    rank, world = 0, int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:                                    # one process per GPU, RCCL over xGMI
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
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
        vgg_weights=args.vgg_weights, compute_dtype=args.dtype, use_hip_graph=args.hip_graph)
    history = trainer.train()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    main(sys.argv[1:])
