"""Worker of tests/test_gpu_dist_trainer.py (launched under torch.distributed.run, 2 ranks sharing one GPU over gloo):
every rank runs the PRODUCT's Trainer.step — engine forward / backward on its shard of graphs, the two-bucket
all-reduce launched from inside the backward (node bucket overlapping the edge-MLP backward), fused Adam — and rank 0
writes the resulting parameters."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nmrgnn_amd import parallel, synth  # noqa: E402
from nmrgnn_amd.engine import Engine  # noqa: E402
from nmrgnn_amd.graph import GraphBatch  # noqa: E402
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space  # noqa: E402
from nmrgnn_amd.train import Trainer  # noqa: E402


def shard_of(b, lo, hi):
    a, z = int(b["graph_ptr"][lo]), int(b["graph_ptr"][hi])
    return dict(atoms=b["atoms"][a:z], nlist=b["nlist"][a:z] - a, edges=b["edges"][a:z], inv_degree=b["inv_degree"][a:z],
                graph_ptr=b["graph_ptr"][lo:hi + 1] - a, y=b["y"][a:z], w=b["w"][a:z])


def run(out_path, n_graphs, steps, world, rank, dev):
    # noise and dropout off: the draws are keyed per rank-local edge, a sharded run cannot replay the full batch's
    hp = declare_gnn_space(HyperParameters(atom_feature_size=64, noise=0.0, dropout=False))
    eng = Engine(hp, 10, device=dev, seed=77)
    full = synth.make_batch(n_graphs, 40, 16, 10, 0.1, seed=12)
    lo, hi = parallel.shard_range(n_graphs, rank, world)
    b = shard_of(full, lo, hi)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=dev)
    y, w = torch.from_numpy(b["y"]).to(dev), torch.from_numpy(b["w"]).to(dev)
    tr = Trainer(eng, lr=1e-3)
    losses = []
    for _ in range(steps):
        losses.append(float(tr.step(gb, y, w, total_graphs=n_graphs).cpu()))
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out_path, flat=eng.params.flat.cpu().numpy(), grad=eng.params.grad.cpu().numpy(), losses=np.asarray(losses))


if __name__ == "__main__":
    out_path, n_graphs, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    world, rank, local = parallel.init_distributed()
    dev = torch.device("cuda", 0)
    run(out_path, n_graphs, steps, world, rank, dev)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
