import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from nmrgnn_amd import synth
from nmrgnn_amd.engine import Engine
from nmrgnn_amd.graph import GraphBatch
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
from nmrgnn_amd.train import Trainer
dev = torch.device("cuda", 0)
hp = declare_gnn_space(HyperParameters(**dict(bench.ARCH, atom_feature_size=256)))
eng = Engine(hp, 10, device=dev, seed=1234)
b = synth.make_batch(512, 256, 16, 10, 0.05, seed=42)
gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=dev)
gb.csc()
y = torch.from_numpy(b["y"]).to(dev); w = torch.from_numpy(b["w"]).to(dev)
tr = Trainer(eng, lr=1e-4)
for _ in range(3): tr.step(gb, y, w)
torch.cuda.synchronize()
os.environ["NG_GW_STAMP"] = "1"
tr.step(gb, y, w)
torch.cuda.synchronize()
print("--- inference forward", file=sys.stderr)
eng.forward(gb)
torch.cuda.synchronize()
