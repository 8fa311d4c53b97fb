"""phase trace of the fused edge backward kernel (block 0, thread 0)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda", 0)
tr = torch.zeros(256, dtype=torch.int64, device=dev)
os.environ["NG_EB_TRACE"] = str(tr.data_ptr())
from nmrgnn_amd import synth
from nmrgnn_amd.engine import Engine
from nmrgnn_amd.graph import GraphBatch
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
hp = declare_gnn_space(HyperParameters(atom_feature_size=64, edge_feature_size=3, edge_hidden_size=128))
eng = Engine(hp, 10, device=dev, seed=1)
b = synth.make_batch(512, 256, 16, 10, 0.05, seed=42)
gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=dev)
for _ in range(2):
    pk = eng.forward(gb, training=True, seed=1)
    eng.backward(torch.ones_like(pk))
torch.cuda.synchronize()
t = tr.cpu().numpy(); t = t[t > 0]; d = np.diff(t)
print("per tile: [rest of tile, prefetch issue, dw1, colsum, barrier]")
for i in range(0, min(len(d), 50), 5): print("  ", d[i:i + 5].tolist())
