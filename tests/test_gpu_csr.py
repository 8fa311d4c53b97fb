"""CSR / variable-degree neighbour lists (BASELINE configs[4] "ragged-batch CSR path").

The CSR form is the reference's padded tuple (nmrgnn/library.py:106-117) with the edges == 0 slots dropped — they
contribute exactly 0 through the edge mask (model.py:251,261).  Checked here:
  * CSR == padded generic path BIT FOR BIT where the summation order allows (everything except the edge-MLP weight
    gradients, whose partial sums follow the tile partition of the edge list);
  * CSR vs the float64 oracle on a distance-cutoff graph of the reference's 7lgi fixture (degree 3..40+), the oracle
    fed the same graph padded to the largest degree — forward and every gradient;
  * the GPU cutoff builder vs a host cKDTree builder;
  * the 100-frame synthetic trajectory of configs[4] (frame 0 + N(0, 0.3 A) jitter, seed 7) through the kNN lists and
    through the cutoff / CSR lists, sampled frames against the oracle;
  * edge_feature_size = 64 (model.py:23) through the feature-chunked kernels, padded and CSR.
Tolerance: 1e-4 on shifts (std in [0.5,2]), 2e-4 of the largest entry on gradients, as tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest

from helpers import make_hp, hp_to_oracle, small_batch, randomize_biases, rel_err

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
PDB2 = os.path.join(HERE, "data", "7lgi.pdb.gz")
PEAK_ATOL, GRAD_RTOL = 1e-4, 2e-4


def csr_to_padded(row_ptr, col, dist):
    """host: the padded (nlist, edges) tuple with K = largest degree, pad = (0, 0.0)"""
    row_ptr = np.asarray(row_ptr, np.int64)
    deg = np.diff(row_ptr)
    N, K = len(deg), max(1, int(deg.max()) if len(deg) else 1)
    nl = np.zeros((N, K), np.int32)
    ed = np.zeros((N, K), np.float32)
    slot = np.arange(len(col)) - np.repeat(row_ptr[:-1], deg)
    rows = np.repeat(np.arange(N), deg)
    nl[rows, slot] = col
    ed[rows, slot] = dist
    return nl, ed


@pytest.mark.parametrize("F", [64, 256])
def test_csr_equals_padded_generic_path_bit_for_bit(gpu_device, monkeypatch, F):
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    monkeypatch.setenv("NG_MP_PATH", "layered")           # the padded any-shape path: same kernels' summation order
    hp = make_hp(atom_feature_size=F, edge_feature_size=3, edge_hidden_size=128)
    b = small_batch(4, 61, seed=3, p_pad=0.2)
    eng = Engine(hp, 10, device=gpu_device, seed=2)
    randomize_biases(eng)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    gc = gb.to_csr()
    assert gc.is_csr and gc.n_edges == int((b["edges"] > 0).sum()) < gb.n_edges
    keep = (b["edges"] > 0).reshape(-1)
    np.testing.assert_array_equal(gc.nlist.cpu().numpy(), b["nlist"].reshape(-1)[keep])
    # inference
    pk_p = eng.forward(gb).cpu().numpy()
    pk_c = eng.forward(gc).cpu().numpy()
    np.testing.assert_array_equal(pk_c, pk_p)
    # training step with the same draws (noise per EDGE: the CSR list takes the kept slots' draws)
    N, K = b["edges"].shape
    xi = eng.randn(N * K, seed=5)
    mask = eng.dropout_mask(N * (F // 2), seed=6)
    dpe = torch.from_numpy(np.random.default_rng(1).standard_normal(N).astype(np.float32)).to(gpu_device)
    tp = eng.forward(gb, training=True, noise=xi, dropout_mask=mask).clone()
    eng.backward(dpe)
    g_p = eng.params.grads_dict()
    xi_c = xi[torch.from_numpy(keep).to(gpu_device)].contiguous()
    tc = eng.forward(gc, training=True, noise=xi_c, dropout_mask=mask).clone()
    eng.backward(dpe)
    g_c = eng.params.grads_dict()
    assert torch.equal(tp, tc)
    for k in g_p:
        if k.startswith("edge_fc/"):
            assert rel_err(g_c[k], g_p[k]) < 1e-5, k        # partial sums follow the tiling of the edge list
        else:
            np.testing.assert_array_equal(g_c[k], g_p[k], err_msg=k)
    # the default padded path (window kernels at F = 64) agrees within tolerance
    monkeypatch.delenv("NG_MP_PATH")
    assert np.max(np.abs(eng.forward(gb).cpu().numpy() - pk_c)) < 5e-5


def _host_cutoff(pos, cutoff, scale=0.1):
    from scipy.spatial import cKDTree
    tree = cKDTree(np.asarray(pos, np.float64))
    lists = tree.query_ball_point(np.asarray(pos, np.float64), cutoff - 1e-9)
    rp, col, dist = [0], [], []
    for i, l in enumerate(lists):
        l = sorted(j for j in l if j != i)
        col += l
        dist += [float(np.linalg.norm(np.asarray(pos[i], np.float64) - np.asarray(pos[j], np.float64))) * scale for j in l]
        rp.append(len(col))
    return np.asarray(rp, np.int64), np.asarray(col, np.int64), np.asarray(dist, np.float64)


@pytest.mark.parametrize("case", ["7lgi", "grid", "tiny"])
def test_cutoff_kernels_give_identical_rows(gpu_device, monkeypatch, case):
    """one wave per atom (default for molecule-sized calls), 16 lanes per atom (NG_KNN=lanes) and one lane per atom
    (NG_KNN=serial): the same CSR rows bit for bit — the count and the fill pass of every form use one distance expression"""
    from nmrgnn_amd.graph import frames_to_batch_cutoff
    from nmrgnn_amd.structure import atoms_onehot, read_pdb
    if case == "7lgi":
        s = read_pdb(PDB2)
        frames, atoms, cut = np.stack(s.frames[:3]), atoms_onehot(s.elements), 4.0
    elif case == "grid":        # integer coordinates: many pairs exactly AT the cutoff distance
        rng = np.random.default_rng(2)
        frames = rng.integers(0, 9, size=(2, 1500, 3)).astype(np.float32)
        atoms, cut = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 1500)], 3.0
    else:
        frames = np.array([[[0, 0, 0], [1, 0, 0], [0, 2, 0], [9, 9, 9]]], np.float32)
        atoms, cut = np.eye(10, dtype=np.float32)[[4, 2, 3, 4]], 2.5
    out = {}
    for mode in ("serial", "lanes", "wave"):
        if mode == "wave":
            monkeypatch.delenv("NG_KNN", raising=False)
        else:
            monkeypatch.setenv("NG_KNN", mode)
        gc = frames_to_batch_cutoff(atoms, frames, cutoff=cut, device=gpu_device)
        out[mode] = [t.cpu().numpy() for t in (gc.row_ptr, gc.nlist, gc.edges, gc.inv_degree, gc.row_of)]
    for mode in ("lanes", "wave"):
        for a, b in zip(out["serial"], out[mode]):
            assert np.array_equal(a, b), mode


def test_cutoff_builder_matches_host(gpu_device):
    from nmrgnn_amd.graph import frames_to_batch_cutoff
    from nmrgnn_amd.structure import atoms_onehot, read_pdb
    s = read_pdb(PDB2)
    frames = np.stack(s.frames[:2])
    atoms = atoms_onehot(s.elements)
    n = atoms.shape[0]
    gc = frames_to_batch_cutoff(atoms, frames, cutoff=4.0, device=gpu_device)
    assert gc.is_csr and gc.N == 2 * n and gc.G == 2
    rp = gc.row_ptr.cpu().numpy().astype(np.int64)
    col, dist, inv = gc.nlist.cpu().numpy(), gc.edges.cpu().numpy(), gc.inv_degree.cpu().numpy()
    deg = np.diff(rp)
    print(f"cutoff 4.0 A on 7lgi: degree min {deg.min()} / median {int(np.median(deg))} / max {deg.max()}, nnz {rp[-1]}")
    assert deg.min() >= 1 and deg.max() >= 30 and deg.max() > 3 * deg.min()          # genuinely variable degree
    for f in range(2):
        hrp, hcol, hdist = _host_cutoff(frames[f], 4.0)
        a, z = rp[f * n], rp[(f + 1) * n]
        # float32 distances right at the cutoff may fall on either side: compare as sets with that slack
        got = set(zip(np.repeat(np.arange(n), deg[f * n:(f + 1) * n]).tolist(), (col[a:z] - f * n).tolist()))
        ref = set(zip(np.repeat(np.arange(n), np.diff(hrp)).tolist(), hcol.tolist()))
        assert len(got ^ ref) <= 4
        if got == ref:
            np.testing.assert_array_equal(col[a:z] - f * n, hcol)                    # ascending neighbour index per row
            np.testing.assert_allclose(dist[a:z], hdist, rtol=3e-6, atol=1e-7)
            cnt = np.zeros(n); np.add.at(cnt, np.repeat(np.arange(n), np.diff(hrp)), hcol > 0)
            np.testing.assert_allclose(inv[f * n:(f + 1) * n], np.where(cnt > 0, 1.0 / np.maximum(cnt, 1), 0.0), rtol=1e-6)
    assert np.all(np.diff(col.astype(np.int64))[np.setdiff1d(np.arange(len(col) - 1), rp[1:-1] - 1)] > 0)


def test_csr_variable_degree_matches_oracle(gpu_device):
    """configs[4]: distance-cutoff 7lgi graph (degree 3..40+) at the baseline width F=256, forward + every gradient
    against the float64 oracle on the same graph padded to the largest degree."""
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import frames_to_batch_cutoff
    from nmrgnn_amd.structure import atoms_onehot, read_pdb
    from oracle import nmrgnn_oracle as O
    s = read_pdb(PDB2)
    atoms = atoms_onehot(s.elements)[:900]
    gc = frames_to_batch_cutoff(atoms, s.frames[0][None, :900], cutoff=4.0, device=gpu_device)
    hp = make_hp(atom_feature_size=256)
    rng = np.random.default_rng(5)
    std = rng.uniform(0.5, 2.0, 10).astype(np.float32)
    avg = rng.uniform(-1.0, 1.0, 10).astype(np.float32)
    eng = Engine(hp, 10, std, avg, device=gpu_device, seed=11)
    sd = randomize_biases(eng)
    N, nnz = gc.N, gc.n_edges
    rp = gc.row_ptr.cpu().numpy()
    deg = np.diff(rp)
    assert deg.max() >= 30 and deg.max() >= 3 * deg.min()           # genuinely variable degree
    nl, ed = csr_to_padded(rp, gc.nlist.cpu().numpy(), gc.edges.cpu().numpy())
    inv = gc.inv_degree.cpu().numpy()
    xi = eng.randn(nnz, seed=123)
    mask = eng.dropout_mask(N * 128, seed=321)
    peaks = eng.forward(gc, training=True, noise=xi, dropout_mask=mask)
    dpe = rng.standard_normal(N).astype(np.float32)
    eng.backward(torch.from_numpy(dpe).to(gpu_device))
    grads = eng.params.grads_dict()
    xi_pad = np.zeros(nl.shape)
    rows = np.repeat(np.arange(N), deg)
    xi_pad[rows, np.arange(nnz) - np.repeat(rp[:-1], deg)] = xi.cpu().numpy()
    ref_pk, ref_g = O.gnn_forward_backward((atoms, nl, ed, inv), sd, hp_to_oracle(hp), dpe, std, avg, training=True,
                                           noise=xi_pad, dropout_mask=(mask.cpu().numpy().reshape(N, 128) > 0))
    assert np.max(np.abs(peaks.cpu().numpy() - ref_pk)) < PEAK_ATOL
    bad = {k: rel_err(grads[k], g) for k, g in ref_g.items() if rel_err(grads[k], g) > GRAD_RTOL}
    assert not bad, bad
    # inference on the same lists
    inf = eng.forward(gc).cpu().numpy()
    assert np.max(np.abs(inf - O.gnn_forward((atoms, nl, ed, inv), sd, hp_to_oracle(hp), std, avg))) < PEAK_ATOL


def test_trajectory_100_frames_knn_and_cutoff(gpu_device):
    """configs[4]: 100 synthetic frames (7lgi frame 0 + N(0, 0.3 A), seed 7) in batches of 25 frames through
    (i) GPU kNN lists (K=16, padded) and (ii) GPU cutoff lists (CSR); sampled frames against the oracle, and the
    batched result against frame-by-frame calls."""
    import nmrgnn_amd
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import frames_to_batch, frames_to_batch_cutoff
    from nmrgnn_amd.structure import atoms_onehot, inv_degree_of, knn_graph, read_pdb
    from oracle import nmrgnn_oracle as O
    s = read_pdb(PDB2)
    rng = np.random.default_rng(7)
    base = s.frames[0]
    frames = np.stack([base + rng.normal(0, 0.3, base.shape).astype(np.float32) for _ in range(100)])
    atoms = atoms_onehot(s.elements)
    n = atoms.shape[0]
    hp = make_hp()                                                   # baseline architecture, F = 256
    r2 = np.random.default_rng(5)
    std = r2.uniform(0.5, 2.0, 10).astype(np.float32)
    avg = r2.uniform(-1.0, 1.0, 10).astype(np.float32)
    eng = Engine(hp, 10, std, avg, device=gpu_device, seed=4)
    sd = randomize_biases(eng)
    ohp = hp_to_oracle(hp)
    knn_pk, cut_pk = [], []
    for b0 in range(0, 100, 25):
        knn_pk.append(eng.forward(frames_to_batch(atoms, frames[b0:b0 + 25], 16, device=gpu_device)).cpu().numpy())
        cut_pk.append(eng.forward(frames_to_batch_cutoff(atoms, frames[b0:b0 + 25], 3.5, device=gpu_device)).cpu().numpy())
    knn_pk = np.concatenate(knn_pk).reshape(100, n)
    cut_pk = np.concatenate(cut_pk).reshape(100, n)
    assert np.mean((knn_pk[-1] - knn_pk[0]) ** 2) > 0                # tests/test_nmrgnn.py:245-257 (frames differ)
    for f in (0, 37, 99):
        # kNN lists: oracle on the host-built graph of the same frame
        nl, ed = knn_graph(frames[f], 16)
        ref = O.gnn_forward((atoms, nl, ed, inv_degree_of(nl)), sd, ohp, std, avg)
        assert np.max(np.abs(knn_pk[f] - ref)) < PEAK_ATOL, f
        # cutoff lists: oracle on the CSR graph padded to its largest degree
        gc = frames_to_batch_cutoff(atoms, frames[f:f + 1], 3.5, device=gpu_device)
        single = eng.forward(gc).cpu().numpy()
        np.testing.assert_allclose(single, cut_pk[f], rtol=0, atol=5e-5)          # batched == frame by frame (other GEMM tiles)
        cnl, ced = csr_to_padded(gc.row_ptr.cpu().numpy(), gc.nlist.cpu().numpy(), gc.edges.cpu().numpy())
        ref = O.gnn_forward((atoms, cnl, ced, gc.inv_degree.cpu().numpy()), sd, ohp, std, avg)
        assert np.max(np.abs(cut_pk[f] - ref)) < PEAK_ATOL, f


@pytest.mark.parametrize("layout", ["padded", "csr"])
def test_mp_layer_edge_feature_size_64(gpu_device, layout):
    """edge_feature_size = 64 (a choice of nmrgnn/model.py:23) through ng_mp_layer_fwd/_bwd (+ _csr) vs NumPy float64."""
    import ctypes as C
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    from nmrgnn_amd.graph import GraphBatch
    rng = np.random.default_rng(64)
    N, K, F, E = 150, 9, 32, 64
    h = rng.standard_normal((N, F)) * 0.5
    nl = rng.integers(0, N, (N, K)).astype(np.int32)
    e = rng.standard_normal((N, K, E)) * 0.2
    pad = rng.random((N, K)) < 0.25
    e[pad] = 0.0
    nl[pad] = 0
    inv = rng.random(N)
    w = rng.standard_normal((F, F, E)) * 0.05
    dH = rng.standard_normal((N, F))
    P = np.einsum("ijn,ijl,lmn,i->im", e, h[nl], w, inv)
    S = np.log1p(np.exp(-np.abs(P))) + np.maximum(P, 0)
    dP = dH / (1 + np.exp(-P)) * inv[:, None]
    dw = np.einsum("ijn,ijl,im->lmn", e, h[nl], dP)
    dA = np.einsum("im,lmn->iln", dP, w)
    de = np.einsum("iln,ijl->ijn", dA, h[nl])
    dh = dH.copy()
    np.add.at(dh, nl.reshape(-1), np.einsum("ijn,iln->ijl", e, dA).reshape(-1, F))
    dev = gpu_device
    t = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    edges = (~pad).astype(np.float32)
    gb = GraphBatch(np.eye(10, dtype=np.float32)[rng.integers(0, 10, N)], nl, edges, inv, device=dev)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    th, tinv, tw, tdH = t(h), t(inv), t(w), t(dH)
    out, tS, A = torch.empty(N, F, device=dev), torch.empty(N, F, device=dev), torch.empty(N, E, F, device=dev)
    tdh, tdw = torch.empty(N, F, device=dev), torch.empty(F, F, E, device=dev)
    if layout == "padded":
        te = t(e)
        csc_ptr, csc_edge = gb.csc()
        tde = torch.zeros(N, K, E, device=dev)
        ctx.check(ctx.lib.ng_mp_layer_fwd(ctx.handle, st, N, K, F, E, 1, 1, ptr(th), ptr(gb.nlist_c), ptr(te), ptr(tinv),
                                          ptr(tw), ptr(out), ptr(A), ptr(tS)), "fwd")
        ctx.check(ctx.lib.ng_mp_layer_bwd(ctx.handle, st, N, K, F, E, 1, ptr(th), ptr(gb.nlist_c), ptr(te), ptr(tinv),
                                          ptr(tw), ptr(A), ptr(tS), ptr(csc_ptr), ptr(csc_edge), ptr(tdH), ptr(tdh),
                                          ptr(tde), 0, ptr(tdw)), "bwd")
        got_de = tde.cpu().numpy()[~pad]
    else:
        gc = gb.to_csr()
        te = t(e[~pad])
        csc_ptr, csc_edge = gc.csc()
        tde = torch.zeros(gc.n_edges, E, device=dev)
        ctx.check(ctx.lib.ng_mp_layer_fwd_csr(ctx.handle, st, N, gc.n_edges, F, E, 1, 1, ptr(th), ptr(gc.row_ptr),
                                              ptr(gc.nlist), ptr(te), ptr(tinv), ptr(tw), ptr(out), ptr(A), ptr(tS)),
                  "fwd")
        ctx.check(ctx.lib.ng_mp_layer_bwd_csr(ctx.handle, st, N, gc.n_edges, F, E, 1, ptr(th), ptr(gc.row_ptr),
                                              ptr(gc.nlist), ptr(gc.row_of), ptr(te), ptr(tinv), ptr(tw), ptr(A), ptr(tS),
                                              ptr(csc_ptr), ptr(csc_edge), ptr(tdH), ptr(tdh), ptr(tde), 0, ptr(tdw)),
                  "bwd")
        got_de = tde.cpu().numpy()
    scale = lambda a: max(1.0, np.abs(a).max())
    np.testing.assert_allclose(out.cpu().numpy(), S + h, rtol=2e-5, atol=2e-5)
    assert np.abs(tdh.cpu().numpy() - dh).max() < 2e-4 * scale(dh)
    assert np.abs(tdw.cpu().numpy() - dw).max() < 2e-4 * scale(dw)
    assert np.abs(got_de - de[~pad]).max() < 2e-4 * scale(de)


def test_from_csr_validation_and_default_inv_degree(gpu_device):
    from nmrgnn_amd.graph import GraphBatch
    atoms = np.eye(10, dtype=np.float32)[[4, 2, 3, 4, 4]]
    # two graphs (3 + 2 atoms); atom 3 (local index 0 of graph 2) is a neighbour of atom 4: not counted in the degree
    row_ptr = [0, 2, 3, 3, 4, 5]
    col = [1, 2, 0, 4, 3]
    dist = [0.1, 0.2, 0.1, 0.15, 0.15]
    gc = GraphBatch.from_csr(atoms, row_ptr, col, dist, graph_ptr=[0, 3, 5], device=gpu_device)
    np.testing.assert_allclose(gc.inv_degree.cpu().numpy(), [0.5, 0.0, 0.0, 1.0, 0.0])   # library.py:115-116 (index 0 never counts)
    np.testing.assert_array_equal(gc.row_of.cpu().numpy(), [0, 0, 1, 3, 4])
    cp, ce = gc.csc()
    np.testing.assert_array_equal(cp.cpu().numpy(), [0, 1, 2, 3, 4, 5])
    np.testing.assert_array_equal(ce.cpu().numpy(), [2, 0, 1, 4, 3])
    with pytest.raises(ValueError):
        GraphBatch.from_csr(atoms, [0, 2, 3, 3, 4, 6], col, dist, device=gpu_device)
    with pytest.raises(ValueError):
        GraphBatch.from_csr(atoms, row_ptr, [1, 2, 0, 4, 7], dist, device=gpu_device)
    with pytest.raises(ValueError):
        GraphBatch.from_csr(atoms, row_ptr, col, [0.1, 0.2, 0.0, 0.15, 0.15], device=gpu_device)


def test_backward_scatter_sum_accepts_incoming_lists_in_any_order(gpu_device):
    """ng_mp_layer_bwd_rec with caller-built incoming-edge lists whose entries do NOT ascend by source (round-5 advisor
    finding: the default-width window pull indexed its staged block with a negative offset for such an entry): same sums."""
    import ctypes as C
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    from nmrgnn_amd.graph import GraphBatch
    rng = np.random.default_rng(21)
    N, K, E, F = 3000, 16, 3, 256
    h = rng.standard_normal((N, F)).astype(np.float32)
    nl = np.clip(np.arange(N)[:, None] + rng.integers(-200, 200, (N, K)), 0, N - 1).astype(np.int32)
    e = rng.random((N, K, E)).astype(np.float32)
    inv = (1.0 / K) * np.ones(N, np.float32)
    w = (rng.standard_normal((F, F, E)) * 0.05).astype(np.float32)
    S = rng.random((N, F)).astype(np.float32)
    dH = rng.standard_normal((N, F)).astype(np.float32)
    gb = GraphBatch(np.eye(10, dtype=np.float32)[rng.integers(0, 10, N)], nl.astype(np.int64), e[:, :, 0], inv, device=gpu_device)
    csc_ptr, csc_edge = gb.csc()
    cp, ce = csc_ptr.cpu().numpy(), csc_edge.cpu().numpy().copy()
    shuf = ce.copy()
    for t in range(N):
        seg = shuf[cp[t]:cp[t + 1]]
        rng.shuffle(seg)
    assert not np.array_equal(shuf, ce)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
    th, tn, te, ti, tw, tS, tdH = t(h), t(nl), t(e), t(inv), t(w), t(S), t(dH)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
    out = []
    for edges in (ce, shuf):
        tce = t(edges.astype(np.int32))
        dh_in = torch.empty(N, F, device=gpu_device)
        de = torch.empty(N * K, E, device=gpu_device)
        dw = torch.empty_like(tw)
        ctx.check(ctx.lib.ng_mp_layer_bwd_rec(ctx.handle, st, N, K, F, E, 1, ptr(th), ptr(tn), ptr(te), ptr(ti), ptr(tw), None, ptr(tS),
                                              ptr(csc_ptr), ptr(tce), ptr(tdH), ptr(dh_in), ptr(de), 0, ptr(dw), None), "bwd")
        torch.cuda.synchronize()
        out.append((dh_in.cpu().numpy(), de.cpu().numpy(), dw.cpu().numpy()))
    for a, b in zip(out[0], out[1]):
        assert np.isfinite(b).all()
        assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max()
