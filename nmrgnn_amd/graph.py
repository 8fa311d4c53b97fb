"""Graph batches in device memory.

The reference feeds ONE graph per model call as the 4-tuple
``(atoms[N,C] one-hot f32, nlist[N,K] int, edges[N,K] f32 distances, inv_degree[N] f32)``
(nmrgnn/library.py:106-117, nmrgnn/model.py:249).  The model is per-atom with indices local to a
graph, so any number of graphs can be concatenated along N with offset neighbour indices
(SURVEY App. C KAT-6); that is how the engine batches molecules.

A GraphBatch additionally carries
  * ``graph_ptr`` [G+1]: atom ranges of the member graphs (loss is per graph, losses.py:37-39);
  * the transposed incoming-edge lists ``csc_ptr`` [N+1] / ``csc_edge`` [nnz] used by the
    deterministic backward scatter (edge id = i*K + j, grouped by target nlist[i,j]); slots with
    ``edges == 0`` are dropped — they carry e == 0 exactly because of the edge mask (model.py:261).
"""
from __future__ import annotations

import numpy as np
import torch


def _to_dev(x, dtype, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    # tf eager tensors and friends expose __array__
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x)), device=device).to(dtype).contiguous()


class GraphBatch:
    def __init__(self, atoms, nlist, edges, inv_degree, graph_ptr=None, device=None,
                 validate=True):
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.atoms = _to_dev(atoms, torch.float32, self.device)
        self.nlist = _to_dev(nlist, torch.int32, self.device)
        self.edges = _to_dev(edges, torch.float32, self.device)
        inv = _to_dev(inv_degree, torch.float32, self.device)
        self.inv_degree = inv.reshape(-1)
        if self.atoms.dim() != 2 or self.nlist.dim() != 2 or self.edges.dim() != 2:
            raise ValueError("atoms must be [N,C], nlist and edges [N,K]")
        self.N, self.C = self.atoms.shape
        self.K = self.nlist.shape[1]
        if self.nlist.shape[0] != self.N or tuple(self.edges.shape) != (self.N, self.K) \
                or self.inv_degree.shape[0] != self.N:
            raise ValueError("inconsistent leading dimensions in graph tuple")
        if validate and self.N > 0:
            lo, hi = int(self.nlist.min()), int(self.nlist.max())
            if lo < 0 or hi >= self.N:
                raise ValueError(f"nlist entries must lie in [0,{self.N}); got [{lo},{hi}]")
        if graph_ptr is None:
            graph_ptr = [0, self.N]
        self.graph_ptr_host = np.asarray(graph_ptr, dtype=np.int32)
        self.graph_ptr = torch.as_tensor(self.graph_ptr_host, device=self.device)
        self.G = len(self.graph_ptr_host) - 1
        self._csc = None
        # compute-side copy of the lists: padded slots (edges == 0, weight exactly 0 after the edge mask)
        # point at the atom itself instead of row 0, so that the row range a tile of atoms references
        # stays local and the window-resident MP kernels (csrc/mp_win.hip) can keep it in LDS
        own = torch.arange(self.N, dtype=torch.int32, device=self.device)[:, None]
        self.nlist_c = torch.where(self.edges > 0, self.nlist, own).contiguous()

    @property
    def n_edges(self):
        return self.N * self.K

    def csc(self):
        """incoming-edge lists for the backward scatter; built once per batch."""
        if self._csc is None:
            N, K = self.N, self.K
            valid = (self.edges > 0).reshape(-1)
            eid = torch.nonzero(valid, as_tuple=False).reshape(-1)
            tgt = self.nlist.reshape(-1)[eid].to(torch.int64)
            order = torch.argsort(tgt, stable=True)
            csc_edge = eid[order].to(torch.int32).contiguous()
            counts = torch.bincount(tgt, minlength=N)
            ptr = torch.zeros(N + 1, dtype=torch.int64, device=self.device)
            ptr[1:] = torch.cumsum(counts, 0)
            self._csc = (ptr.to(torch.int32).contiguous(), csc_edge)
        return self._csc

    def as_tuple(self):
        return self.atoms, self.nlist, self.edges, self.inv_degree


def concat_graphs(graphs, device=None):
    """Concatenate per-graph tuples into one batch, offsetting neighbour indices.
    NB padded slots (nlist == 0, edges == 0) get the offset too; they stay harmless because the
    edge mask zeroes their features and only ``edges > 0`` slots enter the backward lists."""
    atoms, nlist, edges, inv, ptr = [], [], [], [], [0]
    off = 0
    for g in graphs:
        a, nl, e, v = [np.asarray(x) for x in g]
        atoms.append(a.astype(np.float32))
        nlist.append(nl.astype(np.int64) + off)
        edges.append(e.astype(np.float32))
        inv.append(np.asarray(v, np.float32).reshape(-1))
        off += a.shape[0]
        ptr.append(off)
    return GraphBatch(np.concatenate(atoms), np.concatenate(nlist).astype(np.int32),
                      np.concatenate(edges), np.concatenate(inv), graph_ptr=ptr, device=device)


def frames_to_batch(atoms, frames, neighbor_number=16, scale=0.1, device=None):
    """Build the graphs of ``G`` trajectory frames on the GPU (ng_knn_graph) and return them as one
    device-resident GraphBatch: ``atoms`` [n,C] one-hot (shared by all frames), ``frames`` [G,n,3]
    positions in Angstrom.  Same conventions as :func:`nmrgnn_amd.structure.knn_graph`."""
    import ctypes as C
    from . import _lib
    from ._lib import ptr
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    pos = _to_dev(np.asarray(frames, dtype=np.float32) if not isinstance(frames, torch.Tensor) else frames,
                  torch.float32, device)
    if pos.dim() == 2:
        pos = pos[None]
    G, n, _ = pos.shape
    K = int(neighbor_number)
    at = _to_dev(atoms, torch.float32, device)
    if at.shape[0] != n:
        raise ValueError(f"atoms has {at.shape[0]} rows but frames have {n} atoms")
    nlist = torch.empty(G * n, K, dtype=torch.int32, device=device)
    edges = torch.empty(G * n, K, dtype=torch.float32, device=device)
    inv = torch.empty(G * n, dtype=torch.float32, device=device)
    ctx = _lib.get_context(device.index)
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    ctx.check(ctx.lib.ng_knn_graph(ctx.handle, st, G, n, K, float(scale), ptr(pos), ptr(nlist), ptr(edges),
                                   ptr(inv)), "ng_knn_graph")
    ptrs = np.arange(G + 1, dtype=np.int64) * n
    return GraphBatch(at.repeat(G, 1), nlist, edges, inv, graph_ptr=ptrs, device=device, validate=False)
