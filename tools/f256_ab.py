"""A/B of the F=256 training step and its kernels: python tools/f256_ab.py  (env NG_* switches apply)"""
import os, sys, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import bench
from nmrgnn_amd import synth
from nmrgnn_amd.engine import Engine
from nmrgnn_amd.graph import GraphBatch
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
from nmrgnn_amd.train import Trainer
dev = torch.device("cuda", 0)
F = int(os.environ.get("AB_F", "256"))
hp = declare_gnn_space(HyperParameters(**dict(bench.ARCH, atom_feature_size=F)))
eng = Engine(hp, 10, device=dev, seed=1234)
b = synth.make_batch(512, 256, 16, 10, 0.05, seed=42)
gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=dev)
gb.csc()
y = torch.from_numpy(b["y"]).to(dev); w = torch.from_numpy(b["w"]).to(dev)
tr = Trainer(eng, lr=1e-4)
step = lambda: tr.step(gb, y, w)
for _ in range(3): step()
ms = bench.event_timed(step, 10)
prof = bench.profiled_steps(eng, step, 3)
rows = bench.roofline_rows(prof, 3, bench.kernel_work(gb.N, 16, F, 3, 128, 4, 4, 4, 10), h2_gemm=os.environ.get("NG_GEMM_MATH") != "fp32")
print(f"F={F} step median {np.median(ms):.3f} ms  ({gb.N/np.median(ms)/1e3:.2f} M atoms/s)  env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("NG_")))
for r in rows[:16]:
    print("  %-18s x%-3g %7.3f ms/step  %s %s" % (r["kernel"], r["launches_per_step"], r["ms_per_step"], r.get("bound", ""), ("%.3f" % r["frac"]) if "frac" in r else ""))
inf = bench.event_timed(lambda: eng.forward(gb), 5)
print(f"  inference {np.median(inf):.3f} ms")
