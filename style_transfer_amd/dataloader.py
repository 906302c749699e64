"""STDataset / STDataloader with the reference's surface (src/data/dataloader.py:21-170): content and
style image lists paired by index after ``sorted(urls, key=len)``, seeded RandomSampler,
``drop_last=True``, worker seeding, pinned memory; the eval loader yields paths only.

Host-side input pipeline (PIL decode + CPU transforms in DataLoader workers); the HIP hot path
starts at the batch tensors.  Under data parallel every rank gets a disjoint shard of each epoch's
permutation (``rank`` / ``world_size``), so N ranks consume the dataset once per epoch.
"""
from __future__ import annotations

import math
import os
import random
import warnings
from pathlib import Path
from typing import Dict, List

import numpy as np
import PIL.Image
import torch
from PIL import Image, ImageFile
from torch.utils.data import DataLoader, Dataset, Sampler, SequentialSampler

PIL.Image.MAX_IMAGE_PIXELS = 933120000          # dataloader.py:11
ImageFile.LOAD_TRUNCATED_IMAGES = True          # dataloader.py:12


class STDataset(Dataset):
    def __init__(self, content_urls, style_urls, transform_content=None, transform_style=None, url_only: bool = False):
        self.content_urls, self.style_urls = content_urls, style_urls
        self.transform_content, self.transform_style = transform_content, transform_style
        self.url_only = url_only

    def __len__(self) -> int:
        return min(len(self.content_urls), len(self.style_urls))

    @staticmethod
    def _decode(url, transform):
        """One image file -> RGB PIL image -> ``transform`` (if any).  A file PIL cannot read raises IOError after the
        reference's warning (src/data/dataloader.py:45-47 warns and then raises)."""
        try:
            image = Image.open(url).convert('RGB')
        except IOError:
            warnings.warn("IO Error, watch out !")
            raise
        return image if transform is None else transform(image)

    def __getitem__(self, idx) -> Dict:
        pair = {'content_image': (self.content_urls[idx], self.transform_content),
                'style_image': (self.style_urls[idx], self.transform_style)}
        if self.url_only:       # the eval loader: paths only (src/data/dataloader.py:56-58)
            return {key: url for key, (url, _) in pair.items()}
        return {key: self._decode(url, transform) for key, (url, transform) in pair.items()}


class ShardedRandomSampler(Sampler):
    """RandomSampler(generator=g) restricted to rank's slice of each epoch's permutation.  With
    world_size 1 it draws exactly what ``RandomSampler(data_source, generator=g)`` draws, epoch after epoch:
    torch's ``RandomSampler.__iter__`` consumes the generator TWICE per epoch (the second ``randperm`` feeds the
    ``num_samples % n`` tail even when that tail is empty), so a second permutation is drawn and dropped here."""

    def __init__(self, data_source, generator, rank: int = 0, world_size: int = 1):
        self.n, self.generator, self.rank, self.world = len(data_source), generator, rank, world_size

    def __iter__(self):
        perm = torch.randperm(self.n, generator=self.generator).tolist()
        torch.randperm(self.n, generator=self.generator)      # keep the generator in step with RandomSampler
        per = self.n // self.world
        return iter(perm[self.rank * per:(self.rank + 1) * per] if self.world > 1 else perm)

    def __len__(self):
        return self.n // self.world if self.world > 1 else self.n


class STDataloader:
    def __init__(self, content_datapath: List[str], style_datapath: List[str], eval_contentpath: str,
                 eval_stylepath: str, batch_size: int, transform_content, transform_style,
                 max_style_train_samples: int, max_content_train_samples: int, eval_batch_size: int = 1,
                 seed: int = 42, num_worker: int = 2, recursive_paths_load: bool = False,
                 rank: int = 0, world_size: int = 1):
        self.batch_size, self.eval_batch_size = batch_size, eval_batch_size
        self.transform_content, self.transform_style = transform_content, transform_style
        self.recursive_paths_load = recursive_paths_load
        self.max_style_train_samples, self.max_content_train_samples = max_style_train_samples, max_content_train_samples
        self.content_datapath, self.style_datapath = content_datapath, style_datapath
        self.eval_contentpath, self.eval_stylepath = eval_contentpath, eval_stylepath
        self.seed, self.num_worker = seed, num_worker
        self.rank, self.world_size = rank, world_size
        self.generator = torch.Generator()
        self.generator.manual_seed(self.seed)

    @staticmethod
    def get_img_path_list(dir_path, recursive: bool = False) -> List[str]:
        """dataloader.py:92-123: *.jpg / *.png by extension, os.listdir order (or rglob when recursive)."""
        urls = []
        names = os.listdir(dir_path) if not recursive else [str(p) for p in Path(dir_path).rglob("*") if p.is_file()]
        for url in names:
            full = url if recursive else os.path.join(dir_path, url)
            if not os.path.isfile(full):
                warnings.warn("Non file found in data dir")
                continue
            if url.split(".")[-1] in ("jpg", "png"):
                urls.append(full)
            else:
                warnings.warn("non-image file found in dir.")
        return urls

    def __call__(self, *args, **kwargs) -> Dict:
        content_urls: List[str] = []
        per_c = math.floor(self.max_content_train_samples / len(self.content_datapath))
        for path in self.content_datapath:
            print(f"\n Loading {per_c} samples in {path.split('/')[-1]}...")
            content_urls += self.get_img_path_list(path, recursive=self.recursive_paths_load)[:per_c]
        style_urls: List[str] = []
        per_s = math.floor(self.max_style_train_samples / len(self.style_datapath))
        for path in self.style_datapath:
            print(f"\n Loading {per_s} samples in {path.split('/')[-1]}...")
            style_urls += self.get_img_path_list(path, recursive=self.recursive_paths_load)[:per_s]
        content_eval = self.get_img_path_list(self.eval_contentpath) if os.path.isdir(self.eval_contentpath) else []
        style_eval = self.get_img_path_list(self.eval_stylepath) if os.path.isdir(self.eval_stylepath) else []
        train_dataset = STDataset(sorted(content_urls, key=len), sorted(style_urls, key=len),
                                  transform_content=self.transform_content, transform_style=self.transform_style)
        eval_dataset = STDataset(sorted(content_eval, key=len), sorted(style_eval, key=len), url_only=True)
        return {"train": self.get_dataloader(train_dataset, shuffle_flag=True, batch_size=self.batch_size),
                "eval": self.get_dataloader(eval_dataset, shuffle_flag=False, batch_size=self.eval_batch_size)}

    @staticmethod
    def seed_worker(worker_id):
        worker_seed = torch.initial_seed() % 2 ** 32
        np.random.seed(worker_seed)
        random.seed(worker_seed)

    def get_dataloader(self, dataset, shuffle_flag: bool = False, batch_size: int = 1) -> DataLoader:
        sampler = ShardedRandomSampler(dataset, self.generator, self.rank, self.world_size) if shuffle_flag \
            else SequentialSampler(dataset)
        return DataLoader(dataset, sampler=sampler, batch_size=batch_size, drop_last=True,
                          num_workers=self.num_worker, worker_init_fn=self.seed_worker,
                          pin_memory=torch.cuda.is_available())
