"""Training trajectories of the two edge paths side by side: the same model, the same stream of synthetic batches, the same seeds —
per-edge edge MLP against the guarded edge-function table (the Engine's default).  Prints the loss every few steps and the largest
parameter difference at the end.  python tools/train_curve.py [steps] [graphs per batch]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import bench
from nmrgnn_amd import synth
from nmrgnn_amd.engine import Engine
from nmrgnn_amd.graph import GraphBatch
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
from nmrgnn_amd.train import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
graphs = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
hp = declare_gnn_space(HyperParameters(**bench.ARCH))
ea, eb = Engine(hp, bench.NUM_ELEM, device=dev, seed=3), Engine(hp, bench.NUM_ELEM, device=dev, seed=3)
ea.edge_table, eb.edge_table = False, True
eb.edge_table_min_edges = 0
ta, tb = Trainer(ea, lr=1e-3), Trainer(eb, lr=1e-3)
t = lambda a: torch.as_tensor(np.asarray(a)).to(device=dev, dtype=torch.float32)
# a learnable target: the label of an atom is a fixed function of its element and its mean neighbour distance
rng = np.random.default_rng(0)
elem_shift = rng.normal(0, 1, bench.NUM_ELEM).astype(np.float32)
raised = 0
print("step   loss per edge   loss table      |difference|")
for s in range(steps):
    b = synth.make_batch(graphs, 256, bench.K_NEIGH, bench.NUM_ELEM, 0.05, seed=1000 + s)
    d = np.where(b["edges"] > 0, b["edges"], np.nan)
    y = (b["atoms"] @ elem_shift + 2.0 * np.nan_to_num(np.nanmean(d, axis=1) - 0.27)).astype(np.float32)
    raw = (b["atoms"], b["nlist"], b["edges"], b["inv_degree"])
    la = ta.step(GraphBatch(*raw, graph_ptr=b["graph_ptr"], device=dev), t(y), t(b["w"]), seed=s)
    lb = tb.step(GraphBatch(*raw, graph_ptr=b["graph_ptr"], device=dev), t(y), t(b["w"]), seed=s)
    if s % max(1, steps // 10) == 0 or s == steps - 1:
        eb.forward(GraphBatch(*raw, graph_ptr=b["graph_ptr"], device=dev), training=True, seed=s)      # (a taped forward only to read the guard's words)
        rep = eb.edge_table_report()
        eb.tape = None
        print("%4d   %.6f        %.6f        %.2e   %s" % (s, float(la), float(lb), abs(float(la) - float(lb)),
                                                           "" if rep is None else "guard %s err/scale %.2e" % ("UP" if rep[0] else "down", rep[1] / max(rep[2], 1e-30))))
torch.cuda.synchronize()
d = (ea.params.flat - eb.params.flat).abs().max().item()
print("largest parameter difference after %d Adam steps of 1e-3: %.3e (largest parameter %.3f)" % (steps, d, ea.params.flat.abs().max().item()))
