"""which rows of the window gather-GEMM differ from the float64 reference (debug helper for mp_gw.cuh)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
os.environ["NG_MP_GG_MIN_ROWS"] = "1"; os.environ["NG_MP_GG"] = "1"
import test_gpu_mp_gg as T
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
rng = np.random.default_rng(N + 3)
h = rng.standard_normal((N, T.F)).astype(np.float32)
w = (rng.standard_normal((T.F, T.F, 3)) * 0.05).astype(np.float32)
nl, e, inv = T.make_lists(rng, N, 16, 3)
ref, refS = T.ref_fwd(h, nl, e, inv, w)
y1, s1 = T.gpu_fwd(dev, h, nl, e, inv, w)
err = np.abs(y1 - ref).max(axis=1)
bad = np.nonzero(err > 1e-4)[0]
print("N", N, "bad rows", len(bad), "first", bad[:10], "last", bad[-10:])
if len(bad):
    tiles = np.unique(bad // 256)
    print("bad tiles", tiles[:40])
    r = bad[0]
    print("row", r, "bad cols", np.nonzero(np.abs(y1[r] - ref[r]) > 1e-4)[0][:40])
if len(bad):
    r = bad[0]
    P1 = s1[r]; P0 = refS[r]
    print("S ratio", (P1[:6] / P0[:6]).round(4), "res diff", ((y1[r] - s1[r]) - h[r])[:4])
    # which single-row substitution explains it: compare s1[r] against refS of other rows
    d = np.abs(refS - s1[r][None, :]).max(axis=1)
    print("closest ref row", int(np.argmin(d)), float(d.min()))
