"""time the edge forward / backward kernels alone on the bench batch's edge list (2,097,152 edges)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import bench
from nmrgnn_amd.engine import Engine
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
from nmrgnn_amd._lib import ptr, ptr_array
dev = torch.device("cuda", 0)
eng = Engine(declare_gnn_space(HyperParameters(**bench.ARCH)), 10, device=dev, seed=1)
P = eng.params
W = [P[f"edge_fc/{t}/kernel"] for t in range(4)]; B = [P[f"edge_fc/{t}/bias"] for t in range(4)]
dW = [P.g(f"edge_fc/{t}/kernel") for t in range(4)]; dB = [P.g(f"edge_fc/{t}/bias") for t in range(4)]
ne = 2097152
g = torch.Generator(device="cpu").manual_seed(0)
d = (torch.rand(ne, generator=g) * 0.36 + 0.09); d[torch.rand(ne, generator=g) < 0.05] = 0
d = d.to(dev); de = torch.randn(ne, 3, generator=g).to(dev)
e = torch.empty(ne, 3, device=dev); z = torch.empty(3, ne, 128, device=dev)
lib, h = eng.lib, eng.ctx.handle
lay = int(lib.ng_edge_tape_layout(128, 3, 4, 1, ne))
fwd = lambda: eng._ck(lib.ng_edge_mlp_fwd(h, eng._st(), ne, 128, 3, 4, 1, ptr(d), ptr(d), ptr(eng.centers), eng.gap, ptr_array(W), ptr_array(B), ptr(e), ptr(z)), "f")
bwd = lambda: eng._ck(lib.ng_edge_mlp_bwd_tape(h, eng._st(), ne, 128, 3, 4, 1, ptr(d), ptr(d), ptr(eng.centers), eng.gap, ptr_array(W), ptr(z), ptr(de), ptr_array(dW), ptr_array(dB), lay), "b")
fwd_inf = lambda: eng._ck(lib.ng_edge_mlp_fwd(h, eng._st(), ne, 128, 3, 4, 1, ptr(d), ptr(d), ptr(eng.centers), eng.gap, ptr_array(W), ptr_array(B), ptr(e), None), "f")
for _ in range(3): fwd(); bwd(); fwd_inf()
f = np.median(bench.event_timed(fwd, 20)); b = np.median(bench.event_timed(bwd, 20)); fi = np.median(bench.event_timed(fwd_inf, 20))
print("edge fwd %.3f ms (no tape: %.3f; tape 3 x %d x 512 B = %.2f GB -> %.2f TB/s over the difference)  bwd %.3f ms   env: %s"
      % (f, fi, ne, 3 * ne * 512 / 1e9, 3 * ne * 512 / 1e9 / max(f - fi, 1e-9), b, " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("NG_"))))
