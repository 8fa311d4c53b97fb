import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from nmrgnn_amd import _lib
from nmrgnn_amd._lib import ptr
dev = torch.device("cuda", 0); ctx = _lib.get_context(0)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
os.environ["NG_KNN"] = "cells"; _lib.reload_env()
rng = np.random.default_rng(3)
for n in (2770, 110800):
    pos = (rng.random((n, 3)) * (n / 0.1) ** (1 / 3)).astype(np.float32)
    tp = torch.from_numpy(pos).to(dev)
    nl = torch.empty((n, 16), dtype=torch.int32, device=dev); ed = torch.empty((n, 16), device=dev); inv = torch.empty((n,), device=dev)
    run = lambda: ctx.check(ctx.lib.ng_knn_graph(ctx.handle, st, 1, n, 16, 0.1, ptr(tp), ptr(nl), ptr(ed), ptr(inv)), "knn")
    run(); run(); torch.cuda.synchronize()
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(10): run()
    torch.cuda.synchronize()
    print(n, {k: round(v[0] / 10 * 1e3, 1) for k, v in ctx.prof_read().items()}, "us")
    ctx.prof_enable(False)
