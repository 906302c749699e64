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
    """The reference's flags (src/inference.py:22-42: names, short forms, defaults, the start-up checks) plus two."""
    parser = argparse.ArgumentParser(description="Stylise one content image with one style image and save the result")
    add = parser.add_argument
    add('-cpkt', '--path_to_save_cpkt', type=str, help="training checkpoint (transformer*.pth)")
    add('-r', '--res', type=int, default=450, help="short side both images are resized to")
    add('--plot_time', type=int, default=40, help="seconds the comparison figure stays up when shown")
    add('--do_plot', action='store_true', default=True, help="kept for compatibility (always on in the reference)")
    add('--save_dir', type=str, default='./src/output', help="where plot_A{alpha}.png and result_A{alpha}.png go")
    add('-c', '--path_to_content', type=str, help="content image file")
    add('-s', '--path_to_style', type=str, help="style image file")
    add('-a', '--alpha', type=float, default=1.0, help="blend of stylised and content features (1 = fully stylised)")
    # additions (not in the reference)
    add('--encoder_weights', type=str, default=None,
        help="vgg_normalised_conv4_1.pth (default: the reference's cwd-relative path)")
    add('--dtype', choices=["f32", "bf16", "bf16x3"], default="f32", help="conv compute precision")
    args = parser.parse_args(args)
    for path, what in ((args.path_to_save_cpkt, "Please provide the .pth file"),
                       (args.path_to_content, "Please provide the correct content image path"),
                       (args.path_to_style, "Please provide the correct style image path")):
        assert os.path.isfile(path or ""), what
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

    stylised = minmax_to_uint8(result)                    # inference.py:74-76
    comparison = figure_to_image(plot)                   # the 1000 x 400 three-panel figure, inference.py:78-83
    os.makedirs(args.save_dir, exist_ok=True)
    comparison.save(os.path.join(args.save_dir, f"plot_A{args.alpha}.png"))
    stylised.save(os.path.join(args.save_dir, f"result_A{args.alpha}.png"))
    return stylised


def minmax_to_uint8(img: torch.Tensor) -> Image.Image:
    """Normalised decoded image [1,3,H,W] -> 8-bit RGB the way the reference writes ``result_A{alpha}.png``:
    stretch [min, max] onto [0, 1] in float64, scale by 255, truncate."""
    hwc = img.detach().squeeze(0).permute(1, 2, 0).cpu().numpy()
    unit = np.interp(hwc, (float(hwc.min()), float(hwc.max())), (0.0, 1.0))
    return Image.fromarray((unit * 255).astype(np.uint8))


def figure_to_image(plt_module) -> Image.Image:
    """Rasterise the current matplotlib figure (10 x 4 in at 100 dpi) into an RGBA PIL image."""
    fig = plt_module.gcf()
    fig.canvas.draw()
    rgba = np.asarray(fig.canvas.buffer_rgba()).copy()
    plt_module.close('all')
    return Image.fromarray(rgba, mode="RGBA")


if __name__ == "__main__":
    main(sys.argv[1:])
