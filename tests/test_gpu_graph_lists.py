"""ng_build_incoming_lists (csrc/graph_ops.hip): the per-batch incoming-edge lists and the compute-side neighbour list,
built by the library's counting sort, against an independent host construction (stable argsort by target)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host_lists(nlist, edges, N):
    flat_n = nlist.reshape(-1).astype(np.int64)
    live = np.ones(flat_n.shape, bool) if edges is None else (edges.reshape(-1) > 0)
    eid = np.nonzero(live)[0]
    order = np.argsort(flat_n[eid], kind="stable")
    ptr = np.zeros(N + 1, np.int64)
    ptr[1:] = np.cumsum(np.bincount(flat_n[eid], minlength=N))
    return ptr, eid[order]


@pytest.mark.parametrize("graphs,atoms,K,pad", [(1, 5, 2, 0.0), (3, 47, 16, 0.2), (64, 256, 16, 0.05), (2, 1500, 16, 0.5)])
def test_padded_lists_equal_a_stable_sort_by_target(gpu_device, graphs, atoms, K, pad):
    from nmrgnn_amd import synth
    from nmrgnn_amd.graph import GraphBatch
    b = synth.make_batch(graphs, atoms, K, 10, pad, seed=graphs + atoms)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    ptr, eid = gb.csc()
    ptr, eid = ptr.cpu().numpy(), eid.cpu().numpy()
    N = b["nlist"].shape[0]
    hp, he = _host_lists(b["nlist"], b["edges"], N)
    np.testing.assert_array_equal(ptr, hp)
    assert ptr[-1] == len(he)
    np.testing.assert_array_equal(eid[:ptr[-1]], he)
    own = np.arange(N)[:, None]
    np.testing.assert_array_equal(gb.nlist_c.cpu().numpy(), np.where(b["edges"] > 0, b["nlist"], own))
    # run to run: the unordered fill is repaired by the per-target sort
    gb2 = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    assert torch.equal(gb2.csc()[1][:ptr[-1]], gb.csc()[1][:ptr[-1]])


def test_csr_lists_and_a_hub_atom(gpu_device):
    """CSR form (every entry live) with one atom that 3000 others point at (a segment far longer than K)"""
    from nmrgnn_amd.graph import GraphBatch
    rng = np.random.default_rng(5)
    N = 4000
    deg = rng.integers(0, 9, N)
    deg[7] = 0
    row_ptr = np.zeros(N + 1, np.int64)
    row_ptr[1:] = np.cumsum(deg)
    col = rng.integers(0, N, row_ptr[-1]).astype(np.int32)
    col[rng.random(col.shape[0]) < 0.2] = 11          # the hub
    dist = rng.uniform(0.1, 0.4, col.shape[0]).astype(np.float32)
    atoms = np.zeros((N, 10), np.float32)
    atoms[:, 2] = 1
    gb = GraphBatch.from_csr(atoms, row_ptr.astype(np.int32), col, dist, device=gpu_device)
    ptr, eid = gb.csc()
    hp, he = _host_lists(col, None, N)
    np.testing.assert_array_equal(ptr.cpu().numpy(), hp)
    np.testing.assert_array_equal(eid.cpu().numpy()[:len(he)], he)
    assert hp[12] - hp[11] > 2000


def test_scratch_size_and_refusals(gpu_device):
    from nmrgnn_amd import _lib
    ctx = _lib.get_context(0)
    assert ctx.lib.ng_incoming_lists_scratch_bytes(1000, 16000) >= (2 * 1000 + 16000) * 4
    rc = ctx.lib.ng_build_incoming_lists(ctx.handle, None, 10, 4, 39, None, None, None, None, None)
    assert rc != 0


@pytest.mark.gpu
@pytest.mark.parametrize("hub", [40, 64, 65, 700, 4096, 4097, 9000])
def test_incoming_lists_long_segments(gpu_device, hub):
    """segments at and across the thresholds of gl_sort_kernel (16-lane groups up to 64 entries, LDS bitonic sort up to 4096,
    chunked rank sort beyond): lists bit-identical to the host's, and the same from run to run"""
    from nmrgnn_amd.graph import GraphBatch
    rng = np.random.default_rng(hub)
    N = max(2 * hub, 600)
    deg = rng.integers(0, 5, N)
    row_ptr = np.zeros(N + 1, np.int64)
    row_ptr[1:] = np.cumsum(deg)
    col = rng.integers(0, N, row_ptr[-1]).astype(np.int32)
    # exactly `hub` entries point at atom 3 and hub - 1 at atom N - 1 (the last target: its segment end is the list total)
    idx = rng.permutation(col.shape[0])
    col[col == 3] = 5
    col[col == N - 1] = 6
    col[idx[:hub]] = 3
    col[idx[hub:2 * hub - 1]] = N - 1
    dist = rng.uniform(0.1, 0.4, col.shape[0]).astype(np.float32)
    atoms = np.zeros((N, 10), np.float32)
    atoms[:, 2] = 1
    gb = GraphBatch.from_csr(atoms, row_ptr.astype(np.int32), col, dist, device=gpu_device)
    ptr, eid = gb.csc()
    hp, he = _host_lists(col, None, N)
    assert hp[4] - hp[3] == hub and hp[N] - hp[N - 1] == hub - 1
    np.testing.assert_array_equal(ptr.cpu().numpy(), hp)
    np.testing.assert_array_equal(eid.cpu().numpy()[:len(he)], he)
    gb2 = GraphBatch.from_csr(atoms, row_ptr.astype(np.int32), col, dist, device=gpu_device)
    assert torch.equal(gb2.csc()[1][:len(he)], eid[:len(he)])


@pytest.mark.parametrize("atoms,K,pad", [(5, 2, 0.0), (256, 16, 0.05), (700, 16, 0.3), (2048, 8, 0.1)])
def test_one_launch_builds_lists_and_live_view(gpu_device, atoms, K, pad):
    """ng_build_graph_lists (ABI 9): molecule-sized calls build the incoming lists and the live-edge view in ONE launch; the
    outputs are those of the two separate builders, and of the host construction"""
    from nmrgnn_amd import _lib, synth
    from nmrgnn_amd._lib import ptr
    from nmrgnn_amd.graph import GraphBatch
    import ctypes as C
    b = synth.make_batch(1, atoms, K, 10, pad, seed=atoms + K)
    ctx = _lib.get_context(gpu_device.index)
    assert ctx.lib.ng_graph_lists_one_launch(atoms, K) == 1
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    assert gb._live is not None                      # built by the same launch
    perm, pos, d_c, n_live = gb._live
    ne = atoms * K
    p2 = torch.empty(ne, dtype=torch.int32, device=gpu_device)
    q2 = torch.empty(ne, dtype=torch.int32, device=gpu_device)
    d2 = torch.empty(ne, dtype=torch.float32, device=gpu_device)
    n2 = torch.empty(1, dtype=torch.int32, device=gpu_device)
    st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
    ctx.check(ctx.lib.ng_build_live_edges(ctx.handle, st, ne, ptr(gb.edges), ptr(p2), ptr(q2), ptr(d2), ptr(n2)), "live")
    torch.cuda.synchronize()
    nl = int(n_live)
    assert nl == int(n2) == int((b["edges"] > 0).sum())
    assert torch.equal(perm, p2) and torch.equal(pos, q2) and torch.equal(d_c[:nl], d2[:nl])
    cp, ce = gb.csc()
    hp, he = _host_lists(b["nlist"], b["edges"], atoms)
    np.testing.assert_array_equal(cp.cpu().numpy(), hp)
    np.testing.assert_array_equal(ce.cpu().numpy()[:len(he)], he)
    own = np.arange(atoms)[:, None]
    np.testing.assert_array_equal(gb.nlist_c.cpu().numpy(), np.where(b["edges"] > 0, b["nlist"], own))
    assert ctx.lib.ng_graph_lists_one_launch(4096, 16) == 0
