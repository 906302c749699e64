"""Launcher so the reference's command line keeps working from the repo root:
``python src/train.py --output_dir ... --content_datapath ... --style_datapath ...`` (reference src/train.py:176-177)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_amd.train import main  # noqa: E402

if __name__ == "__main__":
    main(sys.argv[1:])
