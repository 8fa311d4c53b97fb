"""micro-benchmark of ng_mp_layer_fwd / _bwd at the bench shape: python tools/mpbench.py [fwd|bwd]"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmrgnn_amd import _lib, synth
from nmrgnn_amd._lib import ptr
from nmrgnn_amd.graph import GraphBatch

dev = torch.device("cuda", 0)
b = synth.make_batch(512, 256, 16, 10, 0.05, seed=42)
gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=dev)
N, K, F, E = gb.N, 16, 64, 3
g = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(N, F, device=dev, generator=g)
e = torch.randn(N, K, E, device=dev, generator=g) * (gb.edges > 0)[..., None]
w = torch.randn(F, F, E, device=dev, generator=g) * 0.1
out = torch.empty(N, F, device=dev); A = torch.empty(N, E, F, device=dev); S = torch.empty(N, F, device=dev)
ctx = _lib.get_context(0)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def fwd():
    ctx.check(ctx.lib.ng_mp_layer_fwd(ctx.handle, st, N, K, F, E, 1, 1, ptr(h), ptr(gb.nlist_c), ptr(e),
                                      ptr(gb.inv_degree), ptr(w), ptr(out), (None if os.environ.get('NO_A') else ptr(A)), ptr(S)), "fwd")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): f()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3
print("%s dbg=%s fwd %.1f us" % (os.environ.get("NG_MP_PATH"), os.environ.get("NG_WIN_DBG"), timeit(fwd)))
