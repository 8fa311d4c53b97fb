"""configs[4]-style measurement: whole-protein inference through the eval-struct driver (baseline
architecture F=256, seeded weights) on tests/data/7lgi.pdb.gz repeated to a 100-frame trajectory."""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nmrgnn_amd
from nmrgnn_amd.graph import frames_to_batch
from nmrgnn_amd.structure import atoms_onehot, read_pdb

warnings.simplefilter("ignore")
s = read_pdb(os.path.join(os.path.dirname(__file__), "..", "tests", "data", "7lgi.pdb.gz"))
rng = np.random.default_rng(7)
frames = np.stack([s.frames[i % len(s.frames)] + rng.normal(0, 0.3, s.frames[0].shape).astype(np.float32)
                   for i in range(100)])
atoms = atoms_onehot(s.elements)
model = nmrgnn_amd.load_model()
model.build(atoms.shape[1])
dev = model.engine.device
for fpb in (1, 10, 50):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); tg = 0.0
        for b0 in range(0, 100, fpb):
            torch.cuda.synchronize()           # the previous batch's model call is asynchronous: do not bill it to the graph build
            t1 = time.perf_counter()
            gb = frames_to_batch(atoms, frames[b0:b0 + fpb], 16, device=dev)
            torch.cuda.synchronize(); tg += time.perf_counter() - t1
            pk = model(gb)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = 100 * atoms.shape[0]
    print(f"frames/batch {fpb:3d}: {dt*1e3:8.1f} ms for 100 frames x {atoms.shape[0]} atoms = {n/dt/1e6:.2f} M atoms/s "
          f"(graph build {tg*1e3:.1f} ms)")

# per-kernel breakdown of one 50-frame batch
eng = model.engine
gb = frames_to_batch(atoms, frames[:50], 16, device=dev)
eng.ctx.prof_reset(); eng.ctx.prof_enable(True)
for _ in range(3):
    model(gb)
torch.cuda.synchronize()
prof = eng.ctx.prof_read(); eng.ctx.prof_enable(False)
print("per-kernel, 50 frames x %d atoms (F=256):" % atoms.shape[0])
for k, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print("  %-18s %8.3f ms/call-set  x%d" % (k, ms / 3, cnt // 3))
