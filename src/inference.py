"""Launcher so the reference's command line keeps working from the repo root:
``python src/inference.py -cpkt X.pth -c C.jpg -s S.jpg -a 1.0`` (reference src/inference.py:94-95)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_amd.inference import main  # noqa: E402

if __name__ == "__main__":
    main(sys.argv[1:])
