"""The multi-process bench flow (what the driver launches for N > 1): two ranks on this box's single GPU
over gloo — barrier, timed steps with the bucketed gradient all-reduce, profile pass on every rank, one
JSON line from rank 0.  (A collective executed by rank 0 only used to dead-lock this flow.)
Also: `bench.py --gpus 2` WITHOUT a torchrun environment must start the two ranks itself, a world size that
differs from --gpus must fail, and strong scaling must shard a fixed total."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _line(res):
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(kw)
    return env


def test_two_rank_bench_line(gpu_device):
    env = _clean_env(NMRGNN_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", BENCH,
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--graphs", "16"]
    d = _line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240))
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "roofline" in d and "cpu_baseline" not in d
    assert d["config"]["backend"] == "gloo" and d["config"]["atoms_total"] == 2 * 16 * 256
    assert d["rank_ms_per_step"]["max"] >= d["rank_ms_per_step"]["min"] > 0 and d["allreduce_exposed_ms"] >= 0


def test_gpus_flag_without_torchrun_starts_the_ranks_itself(gpu_device):
    """`python bench.py --gpus 2` (no torchrun): re-executes under torch.distributed.run with 2 ranks; strong scaling
    over a fixed total of 24 graphs -> 12 per rank"""
    env = _clean_env(NMRGNN_DIST_BACKEND="gloo")
    cmd = [sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--scaling", "strong",
           "--total-graphs", "24", "--no-profile"]
    d = _line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["atoms_total"] == 24 * 256 and d["config"]["atoms_this_rank"] == 12 * 256


def test_eight_rank_strong_scaling_form(gpu_device):
    """what the driver launches on an 8-GPU node (BASELINE configs[3]): `--gpus 8 --scaling strong --total-graphs 4096` under
    torch.distributed.run with EIGHT ranks — here sharing the box's one GPU over gloo, with a small total so that the run stays
    short: shard sizes, one JSON line, exit code 0, every rank through barrier / timed steps / profile pass / destroy"""
    env = _clean_env(NMRGNN_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", "29536", BENCH,
           "--gpus", "8", "--steps", "2", "--warmup", "1", "--scaling", "strong", "--total-graphs", "100"]
    d = _line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600))
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["atoms_total"] == 100 * 256
    # parallel.shard_range(100, 0, 8): the first 100 % 8 = 4 ranks hold 13 graphs, the others 12
    assert d["config"]["atoms_this_rank"] == 13 * 256
    assert d["config"]["world_size"] == 8 and d["config"]["backend"] == "gloo"
    assert d["config"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert d["rank_ms_per_step"]["max"] >= d["rank_ms_per_step"]["min"] > 0


def test_world_size_mismatch_is_an_error(gpu_device):
    env = _clean_env(NMRGNN_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29534", BENCH,
           "--gpus", "4", "--steps", "1", "--warmup", "0", "--graphs", "4"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode != 0
    assert "WORLD_SIZE=2" in res.stderr


def test_nccl_is_required_unless_overridden(gpu_device):
    """two ranks on ONE GPU cannot form an RCCL communicator: without the explicit gloo override the bench refuses"""
    env = _clean_env()
    env.pop("NMRGNN_DIST_BACKEND", None)
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--graphs", "4"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    import torch
    if torch.cuda.device_count() < 2:
        assert res.returncode != 0


def test_single_rank_bench_line_has_all_fields(gpu_device):
    res = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1", "--graphs", "16",
                          "--no-cpu-baseline"], cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=600)
    d = _line(res)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "ms_per_step_hipevent_median", "preprocess_ms",
              "inference", "fp32_mfma_only", "f256", "configs4"):
        assert k in d, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert "error" not in d["f256"] and "error" not in d["configs4"], (d["f256"], d["configs4"])
    assert d["f256"]["value"] > 0 and d["configs4"]["knn16_padded_50_frames_per_call"]["value"] > 0
