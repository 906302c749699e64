"""Side streams that REALLY run next to each other.

A process has GPU_MAX_HW_QUEUES (4 on this pool) hardware queues; HIP deals streams onto them round-robin in the order
the streams are first used, and two streams on one queue run back to back.  Which queue a new stream gets therefore
depends on every stream the process created before (torch's graph-capture streams, other modules' side streams ...):
measured on the Gram-loss step, the same three "lanes" took 47.1 ms or 54.5 ms per step depending on how many
unrelated streams existed.  ``concurrent_lanes`` does not guess: it creates a few candidate streams, times a short
spin kernel on pairs of them, and returns streams that were seen to overlap with the caller's stream and with each
other.  One-time cost: a few milliseconds per device.
"""
from __future__ import annotations

import time
from typing import Dict, List, Tuple

import torch

_CACHE: Dict[Tuple[int, int, int], List[torch.cuda.Stream]] = {}
CANDIDATES = 8
SPIN_CYCLES = 400_000          # a few hundred microseconds


def _pair_ms(a: torch.cuda.Stream, b: torch.cuda.Stream) -> float:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(SPIN_CYCLES)
    if b is not a:
        with torch.cuda.stream(b):
            torch.cuda._sleep(SPIN_CYCLES)
    a.synchronize()
    b.synchronize()
    return (time.perf_counter() - t0) * 1e3


def concurrent_lanes(n: int, main: torch.cuda.Stream = None, priorities=None) -> List[torch.cuda.Stream]:
    """``n`` side streams of the current device that overlap with ``main`` (default: the current stream) and with each
    other, as far as the hardware queues allow (fewer distinct queues than asked for: the list is filled up with
    streams that share one).  ``priorities``: one torch stream priority per lane (-1 high, 0 normal; default all
    normal).  Cached per (device, main stream, n, priorities).  Must not be called under stream capture."""
    main = main or torch.cuda.current_stream()
    prios = tuple(priorities) if priorities is not None else (0,) * n
    assert len(prios) == n
    key = (torch.cuda.current_device(), main.cuda_stream, n, prios)
    if key in _CACHE:
        return _CACHE[key]
    _pair_ms(main, main)                                            # warm-up (first launch of the spin kernel)
    single = min(_pair_ms(main, main) for _ in range(3))
    overlaps = lambda a, b: min(_pair_ms(a, b) for _ in range(2)) < 1.5 * single      # noqa: E731  (serial: 2 x single)
    lanes: List[torch.cuda.Stream] = []
    for prio in prios:
        cands = [torch.cuda.Stream(priority=prio) for _ in range(CANDIDATES)]
        pick = next((c for c in cands if overlaps(c, main) and all(overlaps(c, l) for l in lanes)), None)
        lanes.append(pick if pick is not None else cands[0])        # no free queue left: share one
    _CACHE[key] = lanes
    return lanes
