"""single-frame whole-protein inference (7lgi, 2770 atoms, F=256): eager launches vs one HIP-graph replay"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import bench
from nmrgnn_amd.engine import Engine
from nmrgnn_amd.graph import frames_to_batch, GraphBatch
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
from nmrgnn_amd.structure import atoms_onehot, read_pdb
dev = torch.device("cuda", 0)
s = read_pdb(os.path.join(R, "tests", "data", "7lgi.pdb.gz"))
atoms = torch.from_numpy(atoms_onehot(s.elements)).to(dev)
pos = torch.from_numpy(np.stack(s.frames)).to(dev)
eng = Engine(declare_gnn_space(HyperParameters()), atoms.shape[1], device=dev, seed=1)
gb = frames_to_batch(atoms, pos[:1], 16, device=dev)
for _ in range(3): pk = eng.forward(gb)
torch.cuda.synchronize()
ms = bench.event_timed(lambda: eng.forward(gb), 50)
print("eager model forward, resident graph: median %.3f ms  min %.3f" % (np.median(ms), np.min(ms)))
eng.ctx.prof_reset(); eng.ctx.prof_enable(True)
for _ in range(5): eng.forward(gb)
torch.cuda.synchronize()
prof = eng.ctx.prof_read(); eng.ctx.prof_enable(False)
tot = 0
for k, (t, c) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print("   %-18s x%-2d %.1f us each" % (k, c // 5, t / c * 1e3)); tot += t / 5
print("   sum of kernels %.3f ms" % tot)
# graph capture of the model forward on static buffers
eng.lib.ng_ctx_reserve(eng.ctx.handle, 1 << 28)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(2): eng.forward(gb)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    out = eng.forward(gb)
torch.cuda.synchronize()
def replay(): g.replay()
for _ in range(3): replay()
torch.cuda.synchronize()
ms = bench.event_timed(replay, 50)
print("graph replay: median %.3f ms  min %.3f   same result: %s" % (np.median(ms), np.min(ms), torch.equal(out, pk)))
t0 = time.perf_counter()
for _ in range(100): replay()
torch.cuda.synchronize()
print("100 replays wall %.3f ms each" % ((time.perf_counter() - t0) * 10))
t0 = time.perf_counter()
for _ in range(100): eng.forward(gb)
torch.cuda.synchronize()
print("100 eager forwards wall %.3f ms each" % ((time.perf_counter() - t0) * 10))
