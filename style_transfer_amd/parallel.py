"""Process-group set-up for the data-parallel hot path: one process per GPU, ``torch.distributed`` backend
"nccl" (= RCCL over xGMI on ROCm).  The reference has no distributed code (src/models/trainer.py:103 picks one
device); SURVEY.md 8e is the contract: inference shards images with no collective, training all-reduces one flat
gradient bucket plus the two Hellinger sums per step.

``rehearsal=True`` is for a 1-GPU box: every rank runs on cuda:0 and talks over gloo with the collectives staged
through host memory -- the same collectives in the same order as the RCCL run, so the control flow of N > 1
(no rank-dependent branches, matching collective sequences, exact data-parallel arithmetic) can be tested where
only one GPU exists.  It says nothing about xGMI bandwidth.
"""
from __future__ import annotations

import os

import torch


def init_process_group(rehearsal: bool = False, local_rank: int = None, **kw):
    """Initialise the default group from the launcher's environment (RANK / WORLD_SIZE / MASTER_*) or from
    explicit ``init_method`` / ``rank`` / ``world_size`` keywords; returns (torch.distributed, device)."""
    import torch.distributed as dist
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = 0 if rehearsal else local_rank
    if dev_index >= torch.cuda.device_count():
        raise RuntimeError(f"rank needs GPU {dev_index} but this node shows {torch.cuda.device_count()} "
                           "(rehearsal=True runs every rank on cuda:0 over gloo)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if rehearsal:
        dist.init_process_group("gloo", **kw)
        # gloo's CUDA path opens a stream per collective; with several ranks on one GPU and 4 hardware queues per
        # process those streams can wait on each other forever.  The rehearsal checks control flow and arithmetic,
        # so stage through the host.
        if not getattr(dist.all_reduce, "_st_host_staged", False):
            _all_reduce = dist.all_reduce

            def _host_all_reduce(t, op=dist.ReduceOp.SUM, **kw2):
                if t.is_cuda:
                    c = t.detach().cpu()
                    _all_reduce(c, op=op)
                    t.copy_(c)
                else:
                    _all_reduce(t, op=op)
            _host_all_reduce._st_host_staged = True
            dist.all_reduce = _host_all_reduce
    else:
        dist.init_process_group("nccl", device_id=dev, **kw)     # RCCL over xGMI
    return dist, dev
