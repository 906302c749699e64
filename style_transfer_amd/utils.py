"""Host glue with the reference's names (src/utils/utils.py:22-91): the ``--optim_name`` argparse
action, the wall-clock decorator, seeding and cache clearing."""
from __future__ import annotations

import argparse
import gc
import os
import random
import time
from functools import wraps

import numpy as np
import torch


class ParseKwargsOptim(argparse.Action):
    """``--optim_name adamax betas=0.9,0.99`` -> {'optim_name': 'adamax', 'betas': (0.9, 0.99)}
    (utils.py:22-61): the name is ``optim_name=...`` or the first token without '='; values are
    parsed as int, then float, then a tuple of floats, else kept as strings."""

    def __call__(self, parser, namespace, values, option_string=None):
        out = {}
        named = next((v for v in values if v.startswith("optim_name=")), None)
        if named:
            out["optim_name"] = named.split("=")[1]
        else:
            bare = [v for v in values if "=" not in v]
            if not bare:
                raise ValueError("Please provide the optimizer name!")
            out["optim_name"] = bare[0]
        for item in values:
            if "=" not in item:
                continue
            key, value = item.split("=")
            if key == "optim_name":
                continue
            for cast in (int, float, lambda s: tuple(map(float, s.split(",")))):
                try:
                    value = cast(value)
                    break
                except ValueError:
                    continue
            out[key] = value
        setattr(namespace, self.dest, out)


def timeit(func):
    @wraps(func)
    def timeit_wrapper(*args, **kwargs):
        start = time.perf_counter()
        result = func(*args, **kwargs)
        print(f'Function {func.__name__} Took {time.perf_counter() - start:.4f} seconds')
        return result
    return timeit_wrapper


def set_seed(value):
    print("\n Random Seed: ", value)
    os.environ["CUBLAS_WORKSPACE_CONFIG"] = ":4096:8"
    random.seed(value)
    torch.manual_seed(value)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(value)
    torch.use_deterministic_algorithms(True, warn_only=True)
    np.random.seed(value)


def clear_cuda_cache():
    print("\n Running garbage collection...")
    gc.collect()
    print("\n Clearing cuda cache...")
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
