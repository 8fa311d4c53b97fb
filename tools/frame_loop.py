"""one call per frame: GPU graph build + model forward for 100 frames of 7lgi (what eval-struct does with --frames-per-batch 1)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from nmrgnn_amd.engine import Engine
from nmrgnn_amd.graph import frames_to_batch
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
from nmrgnn_amd.structure import atoms_onehot, read_pdb
dev = torch.device("cuda", 0)
s = read_pdb(os.path.join(R, "tests", "data", "7lgi.pdb.gz"))
rng = np.random.default_rng(7)
frames = np.stack([s.frames[0] + rng.normal(0, 0.3, s.frames[0].shape).astype(np.float32) for _ in range(100)])
atoms = torch.from_numpy(atoms_onehot(s.elements)).to(dev)
pos = torch.from_numpy(frames).to(dev)
eng = Engine(declare_gnn_space(HyperParameters()), atoms.shape[1], device=dev, seed=1)
if hasattr(eng, "freeze_weights") and os.environ.get("FREEZE", "1") == "1":
    eng.freeze_weights(True)
def run():
    for f in range(100):
        eng.forward(frames_to_batch(atoms, pos[f:f + 1], 16, device=dev))
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize()
print("one call per frame: %.3f ms per frame (graph build + model)" % ((time.perf_counter() - t0) * 10))
