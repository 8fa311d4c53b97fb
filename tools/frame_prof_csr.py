"""single-frame whole-protein inference over distance-cutoff CSR lists: per-kernel brackets and wall time"""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, ".")
import nmrgnn_amd
from nmrgnn_amd.graph import frames_to_batch_cutoff
from nmrgnn_amd.structure import atoms_onehot, read_pdb
warnings.simplefilter("ignore")
s = read_pdb("tests/data/7lgi.pdb.gz")
atoms = atoms_onehot(s.elements)
model = nmrgnn_amd.load_model(); model.build(atoms.shape[1]); model.freeze()
eng = model.engine; dev = eng.device
at = torch.from_numpy(atoms).to(dev); pos = torch.from_numpy(np.stack(s.frames[:1])).to(dev)
def frame():
    return model(frames_to_batch_cutoff(at, pos, 3.5, device=dev))
for _ in range(3): frame()
torch.cuda.synchronize()
eng.ctx.prof_reset(); eng.ctx.prof_enable(True)
for _ in range(5): frame()
torch.cuda.synchronize()
prof = eng.ctx.prof_read(); eng.ctx.prof_enable(False)
for k, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print("  %-22s %7.1f us/frame  x%d" % (k, ms / 5 * 1e3, cnt // 5))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): frame()
torch.cuda.synchronize(); print("wall per frame %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
t0 = time.perf_counter()
for _ in range(50): g = frames_to_batch_cutoff(at, pos, 3.5, device=dev)
torch.cuda.synchronize(); print("graph build alone %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): frame()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
