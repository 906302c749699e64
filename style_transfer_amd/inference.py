"""``src/inference.py`` of the reference (argument surface :22-42, main :45-91) on the HIP hot path.

    python -m style_transfer_amd.inference -cpkt X.pth -c content.jpg -s style.jpg [-a 1.0] [-r 450]
        [--save_dir ./src/output] [--plot_time 40]
        [--encoder_weights vgg_normalised_conv4_1.pth] [--dtype f32|bf16]      (additions)

Writes ``plot_A{alpha}.png`` (the 1000x400 three-panel comparison) and ``result_A{alpha}.png`` into
--save_dir, like the reference.  ``--encoder_weights`` exists because the reference hard-codes a
cwd-relative blob (generator.py:78); without the flag that same path is used.
"""
from __future__ import annotations

import argparse
import io
import os
import sys

import numpy as np
import torch
from PIL import Image

from . import precision
from . import transforms as T
from .generator import Decoder, Encoder
from .trainer import Trainer
from .transformer import AdaIN
from .utils import timeit


def parse_args(args):
    parser = argparse.ArgumentParser(description="Inference from the model, plot and save result")
    parser.add_argument('-cpkt', '--path_to_save_cpkt', type=str, help="Path to the save checkpoint")
    parser.add_argument('-r', '--res', type=int, default=450, help="Inference resolution")
    parser.add_argument('--plot_time', type=int, default=40, help="Comparison plot time")
    parser.add_argument('--do_plot', action='store_true', default=True, help="Whether to plot or not")
    parser.add_argument('--save_dir', type=str, default='./src/output', help="Path to save dir")
    parser.add_argument('-c', '--path_to_content', type=str, help="Path to content dir to stylized")
    parser.add_argument('-s', '--path_to_style', type=str, help="Path to style dir to use as style")
    parser.add_argument('-a', '--alpha', type=float, default=1.0, help="alpha value for style and content adjustment")
    # additions (not in the reference)
    parser.add_argument('--encoder_weights', type=str, default=None,
                        help="vgg_normalised_conv4_1.pth (default: the reference's cwd-relative path)")
    parser.add_argument('--dtype', choices=["f32", "bf16", "bf16x3"], default="f32", help="conv compute precision")
    args = parser.parse_args(args)

    assert os.path.isfile(args.path_to_save_cpkt), "Please provide the .pth file"
    assert os.path.isfile(args.path_to_content), "Please provide the correct content image path"
    assert os.path.isfile(args.path_to_style), "Please provide the correct style image path"
    assert 64 < args.res < 1500, "Resolution too small or too large!"
    assert args.plot_time > 1, "Please provide the plot time larger than 0"
    return args


@timeit
def main(args):
    args = parse_args(args)
    device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
    precision.set_compute_dtype(args.dtype)
    encoder = Encoder(weights=args.encoder_weights).to(device)
    decoder = Decoder().to(device)
    checkpoint = torch.load(args.path_to_save_cpkt, map_location="cpu", weights_only=True)
    transformer = AdaIN().to(device)                     # default eps_limit, as inference.py:53
    decoder.load_state_dict(checkpoint['decoder_state_dict'])
    transformer.load_state_dict(checkpoint['model_state_dict'])
    decoder.eval()
    transformer.eval()

    plot, result = Trainer.plot_comparison(
        encoder, decoder, transformer, [args.path_to_content], [args.path_to_style],
        content_transformation=T.Compose([T.Resize(args.res), T.ToTensor(), T.RGBToGrayscaleStacked(False)]),
        style_transformation=T.Compose([T.Resize(args.res), T.ToTensor()]),
        device=device, sleep=args.plot_time, alpha=args.alpha, plot=False)

    image3_np = result.squeeze().permute(1, 2, 0).detach().cpu()
    image3_np = np.interp(image3_np, (image3_np.min(), image3_np.max()), (0, 1))
    result = Image.fromarray((image3_np * 255).astype(np.uint8))

    io_buf = io.BytesIO()
    plot.savefig(io_buf, format='raw')
    plot.close('all')
    io_buf.seek(0)
    result_plot = Image.frombytes('RGBA', (1000, 400), io_buf.getvalue(), 'raw')
    io_buf.close()

    os.makedirs(args.save_dir, exist_ok=True)
    result_plot.save(os.path.join(args.save_dir, f"plot_A{args.alpha}.png"))
    result.save(os.path.join(args.save_dir, f"result_A{args.alpha}.png"))
    return result


if __name__ == "__main__":
    main(sys.argv[1:])
