import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, ".")
import nmrgnn_amd
from nmrgnn_amd.graph import frames_to_batch
from nmrgnn_amd.structure import atoms_onehot, read_pdb
warnings.simplefilter("ignore")
s = read_pdb("tests/data/7lgi.pdb.gz")
atoms = atoms_onehot(s.elements)
model = nmrgnn_amd.load_model(); model.build(atoms.shape[1]); model.freeze()      # eval-struct's use: constant weights
eng = model.engine; dev = eng.device
gb = frames_to_batch(atoms, s.frames[:1], 16, device=dev)
for _ in range(3): model(gb)
eng.ctx.prof_reset(); eng.ctx.prof_enable(True)
for _ in range(5): model(gb)
torch.cuda.synchronize()
prof = eng.ctx.prof_read(); eng.ctx.prof_enable(False)
tot = 0
for k, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print("  %-18s %7.1f us/frame  x%d" % (k, ms / 5 * 1e3, cnt // 5)); tot += ms / 5
print("sum of bracketed kernels %.3f ms" % tot)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): model(gb)
torch.cuda.synchronize(); print("wall per frame %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
