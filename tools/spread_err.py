"""Exploration for the advisor's low finding on mp_win_bwd_node's dw product: upstream gradients spanning 1e-6..1 per atom
(a few labelled atoms, the rest tiny or zero).  Prints, per gradient tensor, the error against the float64 oracle relative to
the tensor's largest entry (the parity tests' measure) and element by element (entries above 1e-3 of the largest)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from helpers import make_hp, hp_to_oracle, small_batch, randomize_biases
from oracle import nmrgnn_oracle as O
from nmrgnn_amd.engine import Engine
from nmrgnn_amd.graph import GraphBatch
from nmrgnn_amd import _lib

dev = torch.device("cuda", 0)
for F in (64, 256):
    hp = make_hp(atom_feature_size=F, edge_feature_size=3, edge_hidden_size=128)
    b = small_batch(6, 120, seed=3)
    rng = np.random.default_rng(5)
    std = rng.uniform(0.5, 2.0, 10).astype(np.float32); avg = rng.uniform(-1, 1, 10).astype(np.float32)
    N = b["atoms"].shape[0]
    dpeaks = (rng.standard_normal(N) * 10.0 ** rng.uniform(-6, 0, N)).astype(np.float32)
    dpeaks[rng.random(N) < 0.5] = 0.0
    res = {}
    for mode in ("f16x2", "fp32"):
        os.environ["NG_GEMM_MATH"] = mode; os.environ["NG_EDGE_MATH"] = mode; _lib.reload_env()
        eng = Engine(hp, 10, std, avg, device=dev, seed=11)
        sd = randomize_biases(eng)
        gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=dev)
        eng.forward(gb, training=True, noise=torch.zeros(N * 16, device=dev), dropout_mask=torch.full((N * F // 2,), 1.25, device=dev))
        eng.backward(torch.from_numpy(dpeaks).to(dev))
        res[mode] = {k: v.copy() for k, v in eng.params.grads_dict().items()}
    _, ref = O.gnn_forward_backward((b["atoms"], b["nlist"], b["edges"], b["inv_degree"]), sd, hp_to_oracle(hp), dpeaks, std, avg,
                                    training=True, noise=np.zeros((N, 16)), dropout_mask=np.ones((N, F // 2)))
    print("F =", F)
    for k in sorted(ref):
        g = np.asarray(ref[k], dtype=np.float64); mx = np.abs(g).max()
        big = np.abs(g) > 1e-3 * mx
        row = [k.ljust(22)]
        for mode in ("f16x2", "fp32"):
            e = np.abs(res[mode][k].astype(np.float64) - g)
            row.append("%s: max/largest %.1e  elementwise %.1e" % (mode, e.max() / mx, (e[big] / np.abs(g[big])).max()))
        print("  ", "   ".join(row))
