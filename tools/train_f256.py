"""Training-step throughput at the reference's DEFAULT architecture (F=256, E=3, H=128) — not the
headline configuration; the MP / FC blocks run the generic layered kernels there."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrgnn_amd import synth
from nmrgnn_amd.engine import Engine
from nmrgnn_amd.graph import GraphBatch
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
from nmrgnn_amd.train import Trainer
dev = torch.device("cuda", 0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hp = declare_gnn_space(HyperParameters(atom_feature_size=F))
eng = Engine(hp, 10, device=dev, seed=1234)
b = synth.make_batch(512, 256, 16, 10, 0.05, seed=42)
gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=dev)
gb.csc()
y = torch.from_numpy(b["y"]).to(dev); w = torch.from_numpy(b["w"]).to(dev)
tr = Trainer(eng, lr=1e-4)
for _ in range(3): tr.step(gb, y, w)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): tr.step(gb, y, w)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"F={F}: {dt*1e3:.2f} ms/step, {gb.N/dt/1e6:.2f} M atoms/s (fwd+bwd+Adam, 512x256 atoms)")
eng.ctx.prof_reset(); eng.ctx.prof_enable(True)
for _ in range(3): tr.step(gb, y, w)
torch.cuda.synchronize()
for k, (ms, cnt) in sorted(eng.ctx.prof_read().items(), key=lambda kv: -kv[1][0])[:12]:
    print("  %-18s %8.3f ms/step x%d" % (k, ms / 3, cnt // 3))
