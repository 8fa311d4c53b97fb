"""The standalone neighbour aggregation (segment-sum, ng_mp_aggregate: A[i,n,l] = sum_j e[i,j,n] h[nlist[i,j],l]) against the
HBM roofline, at the bench batch (F = 64) and at the reference's default width (F = 256).  Algorithmic bytes:
4 (F + K + K E + F E) per atom.  The bench step itself never runs this kernel at F = 64 (the window-resident fused MP
kernels keep the aggregate in LDS); it is the generic path's aggregation and the literal 'scatter-sum' of the north star."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrgnn_amd import _lib, synth
from nmrgnn_amd._lib import ptr
from nmrgnn_amd.graph import GraphBatch

dev = torch.device("cuda", 0)
b = synth.make_batch(512, 256, 16, 10, 0.05, seed=42)
gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=dev)
N, K, E = gb.N, 16, 3
ctx = _lib.get_context(0)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
g = torch.Generator(device=dev).manual_seed(0)
e = torch.randn(N, K, E, device=dev, generator=g) * (gb.edges > 0)[..., None]
import json
out = {"workload": "ng_mp_aggregate on the bench batch: 512 graphs x 256 atoms, K=16, E=3", "rows": []}
for F in (64, 256):
    h = torch.randn(N, F, device=dev, generator=g)
    A = torch.empty(N, E, F, device=dev)
    ctx.check(ctx.lib.ng_ctx_set_graph_span(ctx.handle, 256), "span")    # molecule batch: slab-window kernel at F % 128 == 0
    f = lambda: ctx.check(ctx.lib.ng_mp_aggregate(ctx.handle, st, N, K, F, E, ptr(h), ptr(gb.nlist_c), ptr(e), ptr(A)), "agg")
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20): f()
    t1.record(); torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / 20 * 1e3
    by = 4.0 * N * (F + K + K * E + F * E)
    out["rows"].append({"F": F, "us": us, "algorithmic_MB": by / 1e6, "GBps": by / us / 1e3, "frac_of_8TBps": by / us / 1e3 / 8000,
                        "gathered_MB_through_L2": 4.0 * N * K * F / 1e6})
    print("F=%d: %.1f us, %.0f MB algorithmic -> %.0f GB/s = %.0f %% of 8 TB/s (%.0f %% of the ~6.3 TB/s achievable)" % (
        F, us, by / 1e6, by / us / 1e3, by / us / 1e3 / 80, by / us / 1e3 / 63))

os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/aggbench.json", "w"), indent=1)
