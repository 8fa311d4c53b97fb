#!/usr/bin/env python
"""Micro-benchmark of the neighbour aggregation (the 'scatter-sum' of the hot path):
ng_mp_aggregate at the bench shape, LDS-window kernel vs the plain global-gather kernel,
achieved algorithmic HBM bandwidth (4*(F + K + K*E + F*E) bytes per atom)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nmrgnn_amd import _lib, synth
from nmrgnn_amd._lib import ptr


def run(path, N_graphs=512, F=64, E=3, K=16, iters=50):
    if path:
        os.environ["NG_AGG_PATH"] = path
    else:
        os.environ.pop("NG_AGG_PATH", None)
    dev = torch.device("cuda", 0)
    ctx = _lib.get_context(0)
    b = synth.make_batch(N_graphs, 256, K, 10, 0.05, seed=42)
    N = b["nlist"].shape[0]
    g = torch.Generator(device="cpu").manual_seed(0)
    h = torch.randn(N, F, generator=g).to(dev)
    e = torch.randn(N * K, E, generator=g).to(dev)
    nl = torch.from_numpy(b["nlist"]).to(dev)
    A = torch.empty(N, E, F, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    call = lambda: ctx.check(ctx.lib.ng_mp_aggregate(ctx.handle, st, N, K, F, E, ptr(h), ptr(nl), ptr(e), ptr(A)), "agg")
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        call()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / iters
    alg = 4.0 * N * (F + K + K * E + F * E)
    # reference on the device with torch (plumbing only, as a checker here)
    ref = torch.einsum("ijn,ijl->inl", e.view(N, K, E), h[nl.long()])
    err = float((A - ref).abs().max())
    return dict(path=path or "window", ms=ms, GBs=alg / ms / 1e6, frac_of_8TBs=alg / ms / 1e6 / 8000.0,
                max_abs_err=err, N=N, F=F)


if __name__ == "__main__":
    out = [run(None), run("plain"), run(None, F=256, N_graphs=64), run("plain", F=256, N_graphs=64)]
    for o in out:
        print(json.dumps(o))
