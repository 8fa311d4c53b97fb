"""gl_sort_kernel under rocprofv3: the bench batch's incoming-edge lists (32 builds) and the 3000-entry hub segment of
tests/test_gpu_graph_lists.py.  Run:  rocprofv3 --kernel-trace --stats -d gpurun_out/sort -- python tools/sort_time.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import bench
from nmrgnn_amd.graph import GraphBatch
dev = torch.device("cuda", 0)
from nmrgnn_amd import synth
b = synth.make_batch(bench.GRAPHS_PER_GPU, bench.ATOMS_PER_GRAPH, bench.K_NEIGH, bench.NUM_ELEM, 0.05, seed=42)
raw = [torch.as_tensor(b[k]).to(dev) for k in ("atoms", "nlist", "edges", "inv_degree")]
for _ in range(32):
    gb = GraphBatch(*raw, graph_ptr=b["graph_ptr"], device=dev, validate=False)
    gb.csc()
rng = np.random.default_rng(5)
N = 4000
deg = rng.integers(0, 9, N); deg[7] = 0
row_ptr = np.zeros(N + 1, np.int64); row_ptr[1:] = np.cumsum(deg)
col = rng.integers(0, N, row_ptr[-1]).astype(np.int32)
col[rng.random(col.shape[0]) < 0.2] = 11
dist = rng.uniform(0.1, 0.4, col.shape[0]).astype(np.float32)
atoms = np.zeros((N, 10), np.float32); atoms[:, 2] = 1
for _ in range(8):
    g2 = GraphBatch.from_csr(atoms, row_ptr.astype(np.int32), col, dist, device=dev)
    g2.csc()
torch.cuda.synchronize()
print("done")
