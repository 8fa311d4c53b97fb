"""print the float64 errors of both edge-forward maths for a few sizes (diagnostic; run through gpurun)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from test_gpu_edge_h2 import ref_edge, run_gpu, H

dev = torch.device("cuda:0")
for n, E in [(255, 3), (257, 1), (4096, 3), (70001, 3)]:
    rng = np.random.default_rng(n + E)
    d_src = rng.uniform(0.05, 1.2, n); d_src[rng.random(n) < 0.15] = 0.0
    d_eff = np.where(d_src > 0, d_src + 0.025 * rng.standard_normal(n), d_src)
    centers = np.linspace(0.0, 1.2, H); gap = centers[1] - centers[0]
    Ws = [rng.standard_normal((H, H)) * 0.15 for _ in range(3)] + [rng.standard_normal((H, E)) * 0.2]
    bs = [rng.standard_normal(H) * 0.1 for _ in range(3)] + [rng.standard_normal(E) * 0.1]
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    e_ref, z_ref = ref_edge(f32(d_src), f32(d_eff), f32(centers), float(np.float32(gap)), [f32(w) for w in Ws], [f32(b) for b in bs])
    for math in ("f16x2", "fp32"):
        os.environ["NG_EDGE_MATH"] = math
        from nmrgnn_amd import _lib
        _lib.get_context(0).lib.ng_reload_env()
        e, z = run_gpu(dev, d_src, d_eff, centers, gap, Ws, bs, E, True)
        de = np.abs(e - e_ref)
        i = np.unravel_index(np.argmax(de), de.shape)
        print(n, E, math, "e: max %.2e rms %.2e at %s (ref %.4f) | z max" % (de.max(), np.sqrt((de ** 2).mean()), i, e_ref[i]),
              ["%.2e" % np.abs(z[l] - z_ref[l]).max() for l in range(3)], "z rms", ["%.2e" % np.sqrt(((z[l] - z_ref[l]) ** 2).mean()) for l in range(3)])
