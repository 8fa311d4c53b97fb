"""Trainer.step on two ranks (gloo, sharing this box's GPU) with ENGINE gradients — the HIP forward / backward of each
shard, the node-side all-reduce launched from inside the backward, the edge-side one after it, fused Adam with
grad_scale 1/world — against the single-process step over the whole batch.  (tests/test_dist_gloo.py checks the exchange
arithmetic with oracle gradients on CPU; this one checks the engine + bucket + stream ordering the 8-GPU run uses.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_trainer_worker.py")


@pytest.mark.parametrize("n_graphs", [6, 5])         # even and uneven shards (3 + 2 graphs)
def test_two_rank_trainer_step_equals_the_full_batch_step(gpu_device, tmp_path, n_graphs):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(NMRGNN_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    two = str(tmp_path / "two.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29541 + n_graphs), WORKER, two, str(n_graphs), "3"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    one = str(tmp_path / "one.npz")
    env1 = dict(env)
    res = subprocess.run([sys.executable, WORKER, one, str(n_graphs), "3"], cwd=ROOT, env=env1, capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    a, b = np.load(two), np.load(one)
    scale = np.abs(b["flat"]).max()
    # three Adam steps at lr 1e-3: the sharded sums differ from the full-batch ones by fp32 rounding only
    assert np.abs(a["flat"] - b["flat"]).max() < 2e-5 * scale
    # the exchanged buffer holds the SUM over ranks of the shard-weighted gradients; Adam applies 1/world
    g = np.abs(b["grad"]).max()
    assert np.abs(a["grad"] / 2.0 - b["grad"]).max() < 2e-4 * g
    assert abs(float(b["losses"][-1])) > 0


def test_one_rank_trainer_step_through_rccl_is_the_plain_step(gpu_device, tmp_path):
    """The gradient all-reduces of Trainer.step on the REAL backend ("nccl" = RCCL), in a world of one
    (NMRGNN_FORCE_COLLECTIVES=1: the box has one GPU): async collectives on RCCL's stream launched from inside the backward,
    waited for on the compute stream before the fused Adam.  A sum over one rank is the identity: bit-identical to the step
    without a process group."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MASTER_ADDR="127.0.0.1")
    rccl = str(tmp_path / "rccl.npz")
    env_r = dict(env, NMRGNN_DIST_BACKEND="nccl", NMRGNN_FORCE_COLLECTIVES="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29557", WORKER, rccl, "6", "3"]
    res = subprocess.run(cmd, cwd=ROOT, env=env_r, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    plain = str(tmp_path / "plain.npz")
    res = subprocess.run([sys.executable, WORKER, plain, "6", "3"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    a, b = np.load(rccl), np.load(plain)
    np.testing.assert_array_equal(a["flat"], b["flat"])
    np.testing.assert_array_equal(a["grad"], b["grad"])
    np.testing.assert_array_equal(a["losses"], b["losses"])
