"""Graph-parallel data parallelism: one process per GPU, molecule batches sharded across ranks,
full weight replica per GPU, ONE summing all-reduce of the flat gradient buffer per step
(RCCL over xGMI on GPUs; gloo in the CPU tests).  The reference has no distributed code at all
(SURVEY §2.1); graphs are independent units (no cross-graph edges, per-graph loss
nmrgnn/losses.py:37-39), so the path shards with no activation exchange.

xGMI is point-to-point (7 links x ~153 GB/s per GPU); the gradient bucket is 0.46 MB at F=64 /
4.3 MB at F=256, i.e. latency-bound, so everything goes into at most two calls per step: the
node-side bucket is launched as soon as the MP/FC/head/embedding gradients exist and overlaps with
the edge-MLP backward (whose gradients are produced last); the edge bucket follows.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def force_collectives():
    return os.environ.get("NMRGNN_FORCE_COLLECTIVES") == "1"


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (world, rank, local)."""
    world, rank, local = env_world()
    # NMRGNN_FORCE_COLLECTIVES=1: build the process group and run the gradient all-reduces even in a world of one — a sum
    # over one rank is the identity, so a single-GPU box can exercise the real RCCL path (tests/test_gpu_dist_trainer.py)
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" IS RCCL on ROCm; NMRGNN_DIST_BACKEND=gloo lets several ranks share one GPU (tests)
            backend = os.environ.get("NMRGNN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def shard_range(n_items: int, rank: int, world: int):
    """contiguous, balanced shard [lo, hi) of n_items graphs for this rank"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_grad_weight(local_graphs: int, world: int, total_graphs=None) -> float:
    """Factor for a rank's loss gradient BEFORE the summing all-reduce + 1/world scaling, so that the exchanged
    gradient is the mean over ALL graphs: the local loss is a mean over local_graphs, hence
    (1/world) * weight * g_local = (local_graphs/total_graphs) * g_local.  1.0 for equal shards."""
    if total_graphs is None or world <= 1:
        return 1.0
    return local_graphs * world / float(total_graphs)


class GradBuckets:
    """Splits the flat gradient into the (late) edge bucket and the (early) node bucket."""

    def __init__(self, flat_grad: torch.Tensor, split: int):
        self.flat = flat_grad
        self.split = int(split)
        self.edge = flat_grad[: self.split]
        self.node = flat_grad[self.split:]
        self._pending = []

    def world(self):
        return dist.get_world_size() if dist.is_initialized() else 1

    def rank(self):
        return dist.get_rank() if dist.is_initialized() else 0

    def _exchanging(self):
        return self.world() > 1 or (force_collectives() and dist.is_initialized())

    def launch_node(self):
        if self._exchanging() and self.node.numel():
            self._pending.append(dist.all_reduce(self.node, op=dist.ReduceOp.SUM, async_op=True))

    def launch_edge(self):
        if self._exchanging() and self.edge.numel():
            self._pending.append(dist.all_reduce(self.edge, op=dist.ReduceOp.SUM, async_op=True))

    def wait(self):
        for w in self._pending:
            w.wait()
        self._pending = []

    def grad_scale(self):
        return 1.0 / self.world()
