"""Graph batches in device memory.

The reference feeds ONE graph per model call as the 4-tuple
``(atoms[N,C] one-hot f32, nlist[N,K] int, edges[N,K] f32 distances, inv_degree[N] f32)``
(nmrgnn/library.py:106-117, nmrgnn/model.py:249).  The model is per-atom with indices local to a
graph, so any number of graphs can be concatenated along N with offset neighbour indices
(SURVEY App. C KAT-6); that is how the engine batches molecules.

Besides the reference's padded [N,K] lists a batch can hold the CSR form of SURVEY §8(b)
(``GraphBatch.from_csr`` / ``to_csr`` / ``frames_to_batch_cutoff``): ``row_ptr`` [N+1], ``nlist`` = col [nnz],
``edges`` = dist [nnz] — the padded lists with their ``edges == 0`` slots dropped, or a variable-degree
(distance-cutoff) graph.  ``is_csr`` tells the engine which entry points to call.

A GraphBatch additionally carries
  * ``graph_ptr`` [G+1]: atom ranges of the member graphs (loss is per graph, losses.py:37-39);
  * the transposed incoming-edge lists ``csc_ptr`` [N+1] / ``csc_edge`` [nnz] used by the
    deterministic backward scatter (edge id = i*K + j, grouped by target nlist[i,j]); slots with
    ``edges == 0`` are dropped — they carry e == 0 exactly because of the edge mask (model.py:261).
"""
from __future__ import annotations

import numpy as np
import torch


def _norm_device(device):
    """torch.device with an explicit index for cuda ('cuda' / torch.device('cuda') name the current device): the
    library's per-device context is looked up by index"""
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


_GRAPH_PTR_CACHE = {}


def _graph_ptr_dev(host, device):
    """device copy of a graph_ptr array.  Small partitions repeat from batch to batch (one graph per call: [0, n]; fixed-size
    graphs) and a pageable host-to-device copy blocks the host until the stream has drained — per call that stall was a tenth of
    a one-graph training step — so the copies are kept (at most 64 of at most 4096 entries)."""
    if host.size > 4096 or device.type != "cuda":
        return torch.as_tensor(host, device=device)
    # keyed by the allocating stream as well: a copy made under the prefetcher's side stream is never handed to a batch
    # built on another stream (the caching allocator recycles a block on the stream that allocated it)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, host.tobytes())
    t = _GRAPH_PTR_CACHE.get(key)
    if t is None:
        if len(_GRAPH_PTR_CACHE) >= 64:
            _GRAPH_PTR_CACHE.clear()
        t = torch.as_tensor(host, device=device)
        _GRAPH_PTR_CACHE[key] = t
    return t


def _to_dev(x, dtype, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous()
    # tf eager tensors and friends expose __array__
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x)), device=device).to(dtype).contiguous()


class GraphBatch:
    is_csr = False
    row_ptr = None

    _ctx = None        # library context for the list builders; None = the device's shared one (BatchPrefetcher sets its own)

    def __init__(self, atoms, nlist, edges, inv_degree, graph_ptr=None, device=None,
                 validate=True, nlist_c=None, ctx=None):
        self.device = _norm_device(device)
        self._ctx = ctx
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
        self.graph_ptr = _graph_ptr_dev(self.graph_ptr_host, self.device)
        self.G = len(self.graph_ptr_host) - 1
        self._csc = None
        self._live = None
        # compute-side copy of the lists: padded slots (edges == 0, weight exactly 0 after the edge mask)
        # point at the atom itself instead of row 0, so that the row range a tile of atoms references
        # stays local and the window-resident MP kernels (csrc/mp_win.hip) can keep it in LDS
        if nlist_c is not None:            # the caller knows the lists carry no padded slot (e.g. kNN with n > K)
            self.nlist_c = nlist_c
        else:
            # one library call builds the compute-side list AND the incoming-edge lists (ng_build_incoming_lists)
            self.nlist_c = torch.empty_like(self.nlist)
            self._build_lists(self.nlist_c)

    # ------------------------------------------------------------------ CSR form
    @classmethod
    def from_csr(cls, atoms, row_ptr, col, dist, inv_degree=None, graph_ptr=None, device=None, validate=True, row_of=None):
        """Variable-degree graph(s): row i owns the entries [row_ptr[i], row_ptr[i+1]) of ``col`` (neighbour
        atom, batch-global) and ``dist`` (distance > 0).  ``inv_degree`` defaults to the reference's rule
        1 / #(graph-local neighbour index > 0), 0 when that count is 0 (nmrgnn/library.py:115-116)."""
        self = cls.__new__(cls)
        self.device = _norm_device(device)
        self.is_csr = True
        self.atoms = _to_dev(atoms, torch.float32, self.device)
        if self.atoms.dim() != 2:
            raise ValueError("atoms must be [N,C]")
        self.N, self.C = self.atoms.shape
        self.row_ptr = _to_dev(row_ptr, torch.int32, self.device).reshape(-1)
        self.nlist = _to_dev(col, torch.int32, self.device).reshape(-1)
        self.edges = _to_dev(dist, torch.float32, self.device).reshape(-1)
        self.nnz = int(self.nlist.shape[0])
        self.K = 0
        if self.row_ptr.shape[0] != self.N + 1 or self.edges.shape[0] != self.nnz:
            raise ValueError("row_ptr must be [N+1]; col and dist must have the same length")
        if graph_ptr is None:
            graph_ptr = [0, self.N]
        self.graph_ptr_host = np.asarray(graph_ptr, dtype=np.int32)
        self.graph_ptr = _graph_ptr_dev(self.graph_ptr_host, self.device)
        self.G = len(self.graph_ptr_host) - 1
        if validate:
            rp = self.row_ptr.to(torch.int64)
            if int(rp[0]) != 0 or int(rp[-1]) != self.nnz or bool((rp[1:] < rp[:-1]).any()):
                raise ValueError("row_ptr must start at 0, end at nnz and be non-decreasing")
            if self.nnz:
                lo, hi = int(self.nlist.min()), int(self.nlist.max())
                if lo < 0 or hi >= self.N:
                    raise ValueError(f"col entries must lie in [0,{self.N}); got [{lo},{hi}]")
                if bool((self.edges <= 0).any()):
                    raise ValueError("CSR distances must be > 0 (zero-distance slots are the padded form's mask)")
        if row_of is not None:             # the builder already knows the row of every entry
            self.row_of = _to_dev(row_of, torch.int32, self.device).reshape(-1)
        else:
            deg = (self.row_ptr[1:] - self.row_ptr[:-1]).to(torch.int64)
            self.row_of = torch.repeat_interleave(torch.arange(self.N, device=self.device, dtype=torch.int32),
                                                  deg).contiguous()
        if inv_degree is None:
            gp = torch.as_tensor(self.graph_ptr_host.astype(np.int64), device=self.device)
            rows = self.row_of.to(torch.int64)
            gid = torch.bucketize(rows, gp[1:], right=True)
            local = self.nlist.to(torch.int64) - gp[gid]
            cnt = torch.zeros(self.N, dtype=torch.float32, device=self.device)
            cnt.index_add_(0, rows, (local > 0).to(torch.float32))
            inv_degree = torch.where(cnt > 0, 1.0 / cnt.clamp(min=1.0), torch.zeros_like(cnt))
        self.inv_degree = _to_dev(inv_degree, torch.float32, self.device).reshape(-1)
        if self.inv_degree.shape[0] != self.N:
            raise ValueError("inv_degree must be [N]")
        self.nlist_c = self.nlist
        self._csc = None
        self._live = None
        return self

    def to_csr(self):
        """the same graph(s) with the ``edges == 0`` slots dropped (entry order inside a row is kept)"""
        if self.is_csr:
            return self
        keep = self.edges > 0
        deg = keep.sum(dim=1)
        row_ptr = torch.zeros(self.N + 1, dtype=torch.int64, device=self.device)
        row_ptr[1:] = torch.cumsum(deg, 0)
        return GraphBatch.from_csr(self.atoms, row_ptr.to(torch.int32), self.nlist[keep], self.edges[keep],
                                   self.inv_degree, graph_ptr=self.graph_ptr_host, device=self.device, validate=False)

    @property
    def max_graph_atoms(self):
        """atoms of the largest member graph (a hint for the engine: ng_ctx_set_graph_span)"""
        gp = self.graph_ptr_host
        return int(np.max(np.diff(gp))) if len(gp) > 1 else int(self.N)

    @property
    def n_edges(self):
        return self.nnz if self.is_csr else self.N * self.K

    def _build_lists(self, nlist_c=None):
        """csc_ptr [N+1] / csc_edge [n_entries capacity; csc_ptr[N] live entries] by the library's counting sort — a
        stable sort of the live entries by target, no host synchronisation, no torch kernels (include/nmrgnn_hip.h:
        ng_build_incoming_lists; the reference sees a new graph every step, nmrgnn/library.py:88-89)."""
        if self.device.type != "cuda":
            return self._build_lists_host(nlist_c)
        import ctypes as C
        from . import _lib
        from ._lib import ptr
        ctx = self._ctx or _lib.get_context(self.device.index)
        n_entries = self.n_edges
        csc_ptr = torch.empty(self.N + 1, dtype=torch.int32, device=self.device)
        csc_edge = torch.empty(max(n_entries, 1), dtype=torch.int32, device=self.device)
        if nlist_c is not None and not self.is_csr and self._live is None \
                and ctx.lib.ng_graph_lists_one_launch(self.N, self.K):
            # molecule-sized call: the live-edge view in the same launch (ng_build_graph_lists)
            perm = torch.empty(n_entries, dtype=torch.int32, device=self.device)
            pos = torch.empty(n_entries, dtype=torch.int32, device=self.device)
            d_c = torch.empty(n_entries, dtype=torch.float32, device=self.device)
            n_live = torch.empty(1, dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
                ctx.check(ctx.lib.ng_build_graph_lists(ctx.handle, st, self.N, self.K, ptr(self.nlist), ptr(self.edges),
                                                       ptr(nlist_c), ptr(csc_ptr), ptr(csc_edge), ptr(perm), ptr(pos),
                                                       ptr(d_c), ptr(n_live)), "ng_build_graph_lists")
            self._csc = (csc_ptr, csc_edge)
            self._live = (perm, pos, d_c, n_live)
            return
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            ctx.check(ctx.lib.ng_build_incoming_lists(ctx.handle, st, self.N, 0 if self.is_csr else self.K, n_entries,
                                                      ptr(self.nlist), None if self.is_csr else ptr(self.edges),
                                                      ptr(nlist_c), ptr(csc_ptr), ptr(csc_edge)),
                      "ng_build_incoming_lists")
        self._csc = (csc_ptr, csc_edge)

    def _build_lists_host(self, nlist_c=None):
        """the same lists with torch ops, for batches held in HOST memory (shard bookkeeping in the multi-process CPU
        tests); a device batch never comes here"""
        if nlist_c is not None:
            own = torch.arange(self.N, dtype=torch.int32)[:, None]
            nlist_c.copy_(torch.where(self.edges > 0, self.nlist, own))
        flat = self.nlist.reshape(-1)
        eid = torch.arange(flat.shape[0]) if self.is_csr else torch.nonzero((self.edges > 0).reshape(-1)).reshape(-1)
        tgt = flat[eid].to(torch.int64)
        order = torch.argsort(tgt, stable=True)
        ptr = torch.zeros(self.N + 1, dtype=torch.int64)
        ptr[1:] = torch.cumsum(torch.bincount(tgt, minlength=self.N), 0)
        self._csc = (ptr.to(torch.int32).contiguous(), eid[order].to(torch.int32).contiguous())

    def live_edges(self, force=False):
        """(perm, pos, d_c, n_live) of the padded lists (include/nmrgnn_hip.h: ng_build_live_edges) — the row order of
        the compacted edge kernels; built once per batch on the device, no host synchronisation.  None for a CSR batch
        (every entry is live) and for lists known to carry no padded slot — unless ``force``: the edge-function table's
        guard runs the per-edge kernels over a device-side row count, which only the live view offers (the view of a list
        without dead entries is the identity)."""
        if self.device.type != "cuda" or self.n_edges == 0:
            return None
        if not force and (self.is_csr or self.nlist_c is self.nlist):
            return None
        if self._live is None:
            import ctypes as C
            from . import _lib
            from ._lib import ptr
            ctx = self._ctx or _lib.get_context(self.device.index)
            ne = self.n_edges
            perm = torch.empty(ne, dtype=torch.int32, device=self.device)
            pos = torch.empty(ne, dtype=torch.int32, device=self.device)
            d_c = torch.empty(ne, dtype=torch.float32, device=self.device)
            n_live = torch.empty(1, dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
                ctx.check(ctx.lib.ng_build_live_edges(ctx.handle, st, ne, ptr(self.edges), ptr(perm), ptr(pos), ptr(d_c),
                                                      ptr(n_live)), "ng_build_live_edges")
            self._live = (perm, pos, d_c, n_live)
        return self._live

    def csc(self):
        """incoming-edge lists for the backward scatter; built once per batch.  ``csc_edge`` is allocated for every
        entry; its first ``csc_ptr[N]`` elements are the live ones (the kernels walk it through ``csc_ptr``)."""
        if self._csc is None:
            self._build_lists(None)
        return self._csc

    def as_tuple(self):
        if self.is_csr:
            raise ValueError("a CSR batch has no (atoms, nlist, edges, inv_degree) tuple; see row_ptr / nlist / edges")
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


class BatchPrefetcher:
    """Iterate over graph tuples as device-resident GraphBatches, building batch t+1 while the caller's step t runs.

    The reference sees a new graph tuple every step (nmrgnn/library.py:88-89; keras ``model.fit`` pulls them from a
    ``tf.data`` pipeline with prefetch, nmrgnn/main.py:79-80).  Here the per-batch preprocessing is device work — the copy
    of the tuple, the compute-side lists, the incoming-edge lists and the live-edge view (ng_build_incoming_lists,
    ng_build_live_edges) — a chain of small launches, 0.16 ms for 512 graphs, that the step's kernels would otherwise
    wait behind.  The prefetcher issues that chain on its own HIP stream with its OWN library context (the list builders
    use context scratch; the step's kernels use the shared context's), one batch ahead; the consumer's stream waits on the
    batch's event when it takes the batch.  Results are the same bits as from ``GraphBatch(*tuple)`` on the compute stream.

    ``source`` yields ``(atoms, nlist, edges, inv_degree)`` or ``((atoms, nlist, edges, inv_degree), graph_ptr)``; extra
    keyword arguments go to GraphBatch.  On a CPU device it degenerates to building each batch when it is asked for."""

    def __init__(self, source, device=None, **batch_kw):
        self.source = source
        self.device = _norm_device(device)
        self.batch_kw = batch_kw
        self._stream = None
        self._ctx = None

    def _build(self, item):
        if len(item) == 2 and not hasattr(item[0], "shape"):
            raw, graph_ptr = item
        else:
            raw, graph_ptr = item, None
        if self.device.type != "cuda":
            return GraphBatch(*raw, graph_ptr=graph_ptr, device=self.device, **self.batch_kw), None
        from . import _lib
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
            self._ctx = _lib.Context(self.device.index)
        if any(isinstance(x, torch.Tensor) and x.is_cuda for x in raw):
            # device inputs may have been produced on the consumer's stream: wait for what is enqueued there (at most the
            # previous step).  Host arrays need no such wait — and must not have one: a pageable copy holds the host until it
            # has run, and behind that wait it would run only after the previous step, with the next step not yet enqueued
            self._stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._stream):
            gb = GraphBatch(*raw, graph_ptr=graph_ptr, device=self.device, ctx=self._ctx, **self.batch_kw)
            gb.csc()
            gb.live_edges()
            ready = torch.cuda.Event()
            ready.record(self._stream)
        return gb, ready

    def _hand_over(self, gb, ready):
        if ready is None:
            return gb
        consumer = torch.cuda.current_stream(self.device)
        consumer.wait_event(ready)
        # the tensors were allocated under the side stream: tell the caching allocator who uses them from here on
        # (graph_ptr included: the loss kernels read it on the consumer's stream, and a cached or evicted copy that was
        # allocated under the side stream could otherwise be handed back to the side stream while a step is still queued)
        held = [gb.atoms, gb.nlist, gb.edges, gb.inv_degree, gb.nlist_c, gb.graph_ptr, gb.row_ptr,
                getattr(gb, "row_of", None), *(gb._csc or ()), *(gb._live or ())]
        for t in held:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(consumer)
        gb._ctx = None          # nothing lazy is left to build; later calls on this batch use the shared context
        return gb

    def __iter__(self):
        it = iter(self.source)
        try:
            nxt = self._build(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            try:
                nxt = self._build(next(it))       # issued before the consumer enqueues its step on batch `cur`
            except StopIteration:
                nxt = None
            yield self._hand_over(*cur)


def frames_to_batch(atoms, frames, neighbor_number=16, scale=0.1, device=None):
    """Build the graphs of ``G`` trajectory frames on the GPU (ng_knn_graph) and return them as one
    device-resident GraphBatch: ``atoms`` [n,C] one-hot (shared by all frames), ``frames`` [G,n,3]
    positions in Angstrom.  Same conventions as :func:`nmrgnn_amd.structure.knn_graph`."""
    import ctypes as C
    from . import _lib
    from ._lib import ptr
    device = _norm_device(device)
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
    # n > K: every atom has K real neighbours, no padded slot -> the compute-side list IS the list
    return GraphBatch(at.repeat(G, 1) if G > 1 else at, nlist, edges, inv, graph_ptr=ptrs, device=device, validate=False,
                      nlist_c=nlist if n > K else None)


def frames_to_batch_cutoff(atoms, frames, cutoff=4.0, scale=0.1, device=None):
    """Distance-cutoff graphs of ``G`` trajectory frames, built on the GPU (ng_cutoff_count / ng_cutoff_fill) and
    returned as one device-resident CSR GraphBatch: every other atom of the same frame closer than ``cutoff``
    (Angstrom) is a neighbour, rows in ascending neighbour index, distances x ``scale`` (nm), inv_degree by the
    reference's rule (library.py:115-116).  Variable degree: BASELINE configs[4]."""
    import ctypes as C
    from . import _lib
    from ._lib import ptr
    device = _norm_device(device)
    pos = _to_dev(np.asarray(frames, dtype=np.float32) if not isinstance(frames, torch.Tensor) else frames,
                  torch.float32, device)
    if pos.dim() == 2:
        pos = pos[None]
    G, n, _ = pos.shape
    at = _to_dev(atoms, torch.float32, device)
    if at.shape[0] != n:
        raise ValueError(f"atoms has {at.shape[0]} rows but frames have {n} atoms")
    ctx = _lib.get_context(device.index)
    with torch.cuda.device(device):
        st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        deg = torch.empty(G * n, dtype=torch.int32, device=device)
        ctx.check(ctx.lib.ng_cutoff_count(ctx.handle, st, G, n, float(cutoff), ptr(pos), ptr(deg)), "ng_cutoff_count")
        row_ptr = torch.empty(G * n + 1, dtype=torch.int32, device=device)
        ctx.check(ctx.lib.ng_exclusive_scan_i32(ctx.handle, st, G * n, ptr(deg), ptr(row_ptr)), "ng_exclusive_scan_i32")
        # the one host synchronisation: the list length sizes the buffers.  Summed in int64 — the device scan is int32
        # and a total of 2^32 or more would wrap back to a plausible positive number
        nnz = int(deg.sum(dtype=torch.int64))
        if nnz >= 2 ** 31:
            raise ValueError("cutoff graph: more than 2^31 edges in one batch")
        col = torch.empty(nnz, dtype=torch.int32, device=device)
        dist = torch.empty(nnz, dtype=torch.float32, device=device)
        row_of = torch.empty(nnz, dtype=torch.int32, device=device)
        inv = torch.empty(G * n, dtype=torch.float32, device=device)
        ctx.check(ctx.lib.ng_cutoff_fill_rows(ctx.handle, st, G, n, float(cutoff), float(scale), ptr(pos), ptr(row_ptr),
                                              ptr(col), ptr(dist), ptr(inv), ptr(row_of)), "ng_cutoff_fill_rows")
    ptrs = np.arange(G + 1, dtype=np.int64) * n
    return GraphBatch.from_csr(at.repeat(G, 1) if G > 1 else at, row_ptr, col, dist, inv, graph_ptr=ptrs, device=device,
                               validate=False, row_of=row_of)
