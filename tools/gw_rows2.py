"""debug helper for mp_gw.cuh: only list entry J of every row carries weight; which J are wrong"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
os.environ["NG_MP_GG_MIN_ROWS"] = "1"; os.environ["NG_MP_GG"] = "1"
import test_gpu_mp_gg as T
dev = torch.device("cuda", 0)
N = int(sys.argv[1])
rng = np.random.default_rng(N + 3)
h = rng.standard_normal((N, T.F)).astype(np.float32)
w = (rng.standard_normal((T.F, T.F, 3)) * 0.05).astype(np.float32)
nl, e, inv = T.make_lists(rng, N, 16, 3)
for J in range(16):
    e2 = np.zeros_like(e); e2[:, J, :] = e[:, J, :]
    ref, refS = T.ref_fwd(h, nl, e2, inv, w)
    y1, s1 = T.gpu_fwd(dev, h, nl, e2, inv, w)
    err = np.abs(y1 - ref).max(axis=1)
    bad = np.nonzero(err > 1e-5)[0]
    print("J", J, "bad rows", len(bad), bad[:8], "their sources", nl[bad[:8], J] if len(bad) else "")
