"""per-layer error of the split-operand edge backward vs float64 (debug aid; same setup as tests/test_gpu_edge_h2.py)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import test_gpu_edge_h2 as T
H = T.H
dev = torch.device("cuda", 0)
for n, E in [(1, 3), (32, 3), (33, 3), (64, 3), (65, 3), (128, 3), (200, 3), (5000, 4), (70001, 3)]:
    rng = np.random.default_rng(7 * n + E)
    d_src = rng.uniform(0.05, 1.2, n); d_src[rng.random(n) < 0.15] = 0.0
    d_eff = np.where(d_src > 0, d_src + 0.025 * rng.standard_normal(n), d_src)
    centers = np.linspace(0.0, 1.2, H); gap = centers[1] - centers[0]
    Ws = [rng.standard_normal((H, H)) * 0.15 for _ in range(3)] + [rng.standard_normal((H, E)) * 0.2]
    bs = [rng.standard_normal(H) * 0.1 for _ in range(3)] + [rng.standard_normal(E) * 0.1]
    de = rng.standard_normal((n, E))
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    Wf, bf = [f32(w) for w in Ws], [f32(b) for b in bs]
    _, _, zs = T.ref_edge_bwd(f32(d_src), f32(d_eff), f32(centers), float(np.float32(gap)), Wf, bf, f32(de))
    zs32 = [f32(z) for z in zs]
    m = (f32(d_src) > 0).astype(np.float64)
    Rm = np.exp(-(f32(d_eff)[:, None] - f32(centers)[None, :]) ** 2 / float(np.float32(gap))) * m[:, None]
    xs = [Rm] + zs32
    dE = f32(de) * m[:, None]
    ref_dW, ref_db = [None] * 4, [None] * 4
    ref_dW[3], ref_db[3] = xs[3].T @ dE, dE.sum(0)
    g = dE @ Wf[3].T
    for l in (2, 1, 0):
        G = g * (1.0 - np.exp(-xs[l + 1]))
        ref_dW[l], ref_db[l] = xs[l].T @ G, G.sum(0)
        g = G @ Wf[l].T
    dW, db = T.run_gpu_bwd(dev, d_src, d_eff, centers, gap, Ws, zs32, de, E)
    print(n, E, " ".join("dW%d %.1e db%d %.1e |" % (l, np.abs(dW[l] - ref_dW[l]).max() / max(np.abs(ref_dW[l]).max(), 1e-9), l,
                                                      np.abs(db[l] - ref_db[l]).max() / max(np.abs(ref_db[l]).max(), 1e-9)) for l in range(4)))
