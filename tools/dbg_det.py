import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, ctypes as C
from helpers import make_hp, randomize_biases
from nmrgnn_amd.engine import Engine
from nmrgnn_amd._lib import ptr, ptr_array
dev = torch.device("cuda", 0)
hp = make_hp(atom_feature_size=64)
eng = Engine(hp, 10, device=dev, seed=2); randomize_biases(eng)
P = eng.params
W = [P[f"edge_fc/{t}/kernel"] for t in range(4)]; B = [P[f"edge_fc/{t}/bias"] for t in range(4)]
lib, h = eng.lib, eng.ctx.handle
def run(ne, train, d):
    st = eng._st()
    e = torch.empty(ne, 3, device=dev)
    z = torch.empty(3, ne, 128, device=dev) if train else None
    eng._ck(lib.ng_edge_mlp_fwd(h, st, ne, 128, 3, 4, 1, ptr(d), ptr(d), ptr(eng.centers), eng.gap, ptr_array(W), ptr_array(B), ptr(e), ptr(z)), "f")
    return e, z
rng = np.random.default_rng(0)
for ne in (3159, 3904, 100000, 2097152):
    d = torch.from_numpy(rng.uniform(0.05, 0.5, ne).astype(np.float32)).to(dev)
    for train in (False, True):
        e0, z0 = run(ne, train, d)
        nbad = 0; worst = 0.0; where = []
        for rep in range(30):
            e1, z1 = run(ne, train, d)
            if not torch.equal(e0, e1):
                nbad += 1
                idx = (e0 != e1).any(dim=1).nonzero().reshape(-1)
                worst = max(worst, (e0 - e1).abs().max().item()); where.append((idx.min().item(), idx.max().item(), idx.numel()))
            if train and not torch.equal(z0, z1):
                where.append("z differs")
        print(f"ne={ne} train={train}: {nbad}/30 runs differ from the first, worst {worst:.2e}", where[:6])
