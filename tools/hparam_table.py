"""ms per training step and the kernels taken across the reference's hyper-parameter space (nmrgnn/model.py:15-27:
atom_feature_size in {32,64,128,256}, edge_hidden_size in {16..256}, edge_feature_size in {1,2,3,8,64}) on the bench batch
(512 x 256 atoms, K = 16).  Verdict round 4, item 6:  F=128/E=3/H=128, F=128/E=64/H=64 (the commented "small AMP"
default), F=64/E=8 — next to the two measured widths."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrgnn_amd import synth
from nmrgnn_amd.engine import Engine
from nmrgnn_amd.graph import GraphBatch
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
from nmrgnn_amd.train import Trainer
dev = torch.device("cuda", 0)
b = synth.make_batch(512, 256, 16, 10, 0.05, seed=42)
y = torch.from_numpy(b["y"]).to(dev); w = torch.from_numpy(b["w"]).to(dev)
CONFIGS = [dict(F=64, E=3, H=128), dict(F=256, E=3, H=128), dict(F=128, E=3, H=128), dict(F=128, E=64, H=64), dict(F=64, E=8, H=128),
           dict(F=32, E=2, H=64)]
for c in CONFIGS:
    hp = declare_gnn_space(HyperParameters(atom_feature_size=c["F"], edge_feature_size=c["E"], edge_hidden_size=c["H"]))
    eng = Engine(hp, 10, device=dev, seed=1234)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=dev)
    gb.csc()
    tr = Trainer(eng, lr=1e-4)
    for _ in range(3): tr.step(gb, y, w)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.step(gb, y, w)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    # algorithmic forward flops per atom: edge K((Le-1)H^2 + HE), MP L(KFE + F^2 E), FC (Lf-1)F^2 + F^2/2 MACs
    F, E, H, K = c["F"], c["E"], c["H"], 16
    mac = K * (3 * H * H + H * E) + 4 * (K * F * E + F * F * E) + 3 * F * F + F * F // 2
    print(f"F={F} E={E} H={H}: {dt*1e3:.2f} ms/step  {gb.N/dt/1e6:.2f} M atoms/s   fwd {2*mac/1e6:.2f} MFLOP/atom  "
          f"-> {3*2*mac*gb.N/dt/1e12:.1f} TF fwd+bwd algorithmic")
    eng.ctx.prof_reset(); eng.ctx.prof_enable(True)
    for _ in range(3): tr.step(gb, y, w)
    torch.cuda.synchronize()
    rows = sorted(eng.ctx.prof_read().items(), key=lambda kv: -kv[1][0])
    print("   " + " | ".join("%s %.3f x%d" % (k, ms / 3, cnt // 3) for k, (ms, cnt) in rows[:9]))
    eng.ctx.prof_enable(False)
    del eng, tr, gb
    torch.cuda.empty_cache()
