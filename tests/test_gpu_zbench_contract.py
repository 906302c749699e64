"""The driver's contract on bench.py: ONE JSON line on stdout from rank 0 with the agreed keys, a roofline object for
the dominant kernel and (when asked) a CPU baseline.  Small shapes so that the run takes seconds; collected last."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run(
        [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
         "--side", "128", "--train-batch", "2", "--train-side", "64", "--train-steps", "2", "--train-warmup", "1",
         "--no-gram", "--no-cpu-baseline"],
        capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "img/s" and d["value"] > 0 and d["dtype"] == "f32" and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # executed work next to the algorithmic figure (the sub-pixel upsample layers perform 16/36 of their multiply-adds)
    assert "16 multiply-adds per source pixel" in r["method"]
    cfg = d["config"]
    assert 0 < cfg["executed_gflop_per_image"] < cfg["gflop_per_image"]
    assert 0 < d["model_tflops_executed"] < d["model_tflops"]
    assert abs(d["model_tflops_executed"] / d["model_tflops"] - cfg["executed_gflop_per_image"] / cfg["gflop_per_image"]) < 2e-2
    assert 0 < d["hires"]["model_tflops_executed"] < d["hires"]["model_tflops"]
    assert d["train"]["value"] > 0 and d["bf16"]["inference"]["value"] > 0 and d["bf16x3"]["inference"]["value"] > 0
    for name, bound, unit in (("roofline_hist", "hbm", "GB/s"), ("roofline_tv", "hbm", "GB/s"), ("roofline_cov", "mfma", "TFLOP/s")):
        o = d[name]
        assert o["bound"] == bound and o["unit"] == unit and 0 < o["frac"] < 1 and "traffic" in o, name
        assert abs(o["frac"] - o["achieved"] / o["peak"]) < 1e-3


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (a torch.distributed.run child,
    started before the parent touches the GPU) and forwards rank 0's ONE JSON line.  ST_BENCH_REHEARSAL=1: both ranks
    on cuda:0 over gloo (this box has one GPU); the data-parallel training legs -- eager and the three-graph replay --
    run their real collectives."""
    env = dict(os.environ, ST_BENCH_REHEARSAL="1", ST_BENCH_PREWARM="0.05")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run(
        [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
         "--side", "64", "--train-batch", "2", "--train-side", "64", "--train-steps", "2", "--train-warmup", "1",
         "--no-gram", "--no-bf16"],
        capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["train"]["n_gpus"] == 2 and d["train"]["value"] > 0
    assert "dp2" in d["train"]["config"]["parallelism"]
    assert d["train"]["hip_graph"].get("value", 0) > 0, d["train"]["hip_graph"]
    assert "cpu_baseline" not in d                                   # rank 0 at N = 1 only


def test_bench_under_launcher_on_real_rccl_one_rank():
    """The driver's launch form (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) with N = 1
    and ST_FORCE_DP=1: process group on backend "nccl" (RCCL), barrier / max-over-ranks / both training all-reduces
    and the three-graph replay all go through the real backend on this box's one GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, ST_FORCE_DP="1", ST_BENCH_PREWARM="0.05")
    env.pop("ST_BENCH_REHEARSAL", None)
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "1", "--mode", "train", "--train-batch", "2",
         "--train-side", "64", "--train-steps", "3", "--train-warmup", "1", "--no-gram", "--no-bf16", "--no-cpu-baseline"],
        capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "dp1" in d["config"]["parallelism"]
    assert d["hip_graph"]["value"] > 0 and "three HIP graph" in d["hip_graph"]["launch"], d["hip_graph"]


def test_bench_gpus_2_train_mode_rehearsal_has_no_failed_leg():
    """`python bench.py --gpus 2 --mode train --no-bf16 --no-gram` (what a data-parallel scaling run executes per N, at
    rehearsal sizes): finishes, reports n_gpus 2 from rank 0 alone, and NO leg carries an "error" key (the secondary
    legs are guarded under a launcher: a failure would be recorded instead of raised)."""
    env = dict(os.environ, ST_BENCH_REHEARSAL="1", ST_BENCH_PREWARM="0.05")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run(
        [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--mode", "train", "--no-bf16", "--no-gram",
         "--train-batch", "2", "--train-side", "64", "--train-steps", "3", "--train-warmup", "1"],
        capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["metric"] == "train img/s" and d["value"] > 0 and d["scaling"] == "weak"
    assert "dp2" in d["config"]["parallelism"]

    def errors(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "error":
                    yield path
                yield from errors(v, f"{path}/{k}")
    assert list(errors(d)) == [], list(errors(d))
    assert d["hip_graph"]["value"] > 0 and "three HIP graph" in d["hip_graph"]["launch"]
