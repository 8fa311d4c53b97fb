"""debug helper for mp_gw.cuh: which source row does entry J of each row actually gather"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
os.environ["NG_MP_GG_MIN_ROWS"] = "1"; os.environ["NG_MP_GG"] = "1"
import test_gpu_mp_gg as T
dev = torch.device("cuda", 0)
N = int(sys.argv[1]); J = int(sys.argv[2])
rng = np.random.default_rng(N + 3)
F = T.F
h = np.repeat(np.arange(N, dtype=np.float32)[:, None], F, axis=1) + np.arange(F, dtype=np.float32)[None, :] / 1024.0
w = np.zeros((F, F, 3), np.float32); w[np.arange(F), np.arange(F), 0] = 1.0
nl, e, inv = T.make_lists(rng, N, 16, 3)
e2 = np.zeros_like(e); e2[:, J, 0] = 1.0
inv = np.ones_like(inv)
y1, s1 = T.gpu_fwd(dev, h, nl, e2, inv, w, act=0)
got = s1            # = h[src]
exp = h[nl[:, J]]
bad = np.nonzero(np.abs(got - exp).max(axis=1) > 1e-3)[0]
print("bad rows", len(bad))
for r in bad[:12]:
    g = got[r]
    print("row", r, "expected src", nl[r, J], "got row-ish", np.round(g[:4], 3), "cols wrong", int((np.abs(got[r] - exp[r]) > 1e-3).sum()),
          "nl row", nl[r].tolist())
