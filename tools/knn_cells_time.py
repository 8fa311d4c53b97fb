"""ng_knn_graph on large frames: brute force (NG_KNN=brute) against the cell grid (knn_cells.hip), hipEvent-timed.
Point sets: a uniform box at 0.1 atoms per cubic Angstrom, and 7lgi tiled to the same atom counts (protein density inside
the copies, empty space between them)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrgnn_amd import _lib  # noqa: E402
from nmrgnn_amd._lib import ptr  # noqa: E402
from nmrgnn_amd.structure import read_pdb  # noqa: E402

dev = torch.device("cuda", 0)
ctx = _lib.get_context(0)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
prot = np.asarray(read_pdb("tests/data/7lgi.pdb.gz").frames[0], dtype=np.float32)
ext = prot.max(0) - prot.min(0) + 9.0


def tiled(copies):
    side = int(np.ceil(copies ** 0.5))
    t = [prot + np.array([(c % side) * ext[0], (c // side) * ext[1], 0.0], dtype=np.float32) for c in range(copies)]
    return np.concatenate(t)


def timed(pos, mode, reps):
    os.environ["NG_KNN"] = mode
    _lib.reload_env()
    n = pos.shape[0]
    tp = torch.from_numpy(pos).to(dev)
    nl = torch.empty((n, 16), dtype=torch.int32, device=dev)
    ed = torch.empty((n, 16), device=dev)
    inv = torch.empty((n,), device=dev)
    run = lambda: ctx.check(ctx.lib.ng_knn_graph(ctx.handle, st, 1, n, 16, 0.1, ptr(tp), ptr(nl), ptr(ed), ptr(inv)), "knn")
    run(); run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, nl


rng = np.random.default_rng(3)
for n in (2770, 11080, 33240, 110800, 443200):
    for kind in ("tiled 7lgi", "uniform box"):
        pos = tiled(n // 2770) if kind == "tiled 7lgi" else (rng.random((n, 3)) * (n / 0.1) ** (1 / 3)).astype(np.float32)
        tb, a = timed(pos, "brute", 3 if n > 100000 else 10)
        tc, b = timed(pos, "cells", 10)
        print(f"{kind:12s} n = {n:7d}: brute force {tb:9.3f} ms   cell grid {tc:7.3f} ms   x{tb / tc:7.1f}   same lists: {bool(torch.equal(a, b))}")
