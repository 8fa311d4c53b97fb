"""The sixteen-wave window kernels (round 4: csrc/mp_win16.hip forward, mp_win16_bwd.hip edge-side backward — the defaults
for atom_feature_size 64 on calls the round-6 wave-autonomous forward does not take) against the eight-wave kernels they
replace (NG_MP_W16=0), through the whole engine: peaks and every gradient.  (The sixteen-wave node-side kernel, measured
10 % slower than the eight-wave one, left the library in round 6: tools/variants/mp_win16_node.hip.)
  forward        same per-atom gather, the matrix sums meet in another order: agreement to fp32 rounding
  edge backward  the same products and dots in the same order: every gradient bit for bit
Shapes: a tail tile, graphs smaller than a tile, one graph spanning many tiles, K < 16, E = 1 and 2."""
import numpy as np
import pytest

from helpers import make_hp, small_batch

pytestmark = pytest.mark.gpu


def _run(dev, b, E, seed=3):
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    eng = Engine(make_hp(atom_feature_size=64, edge_feature_size=E), 10, device=dev, seed=seed)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=dev)
    N, K = gb.edges.shape
    inf = eng.forward(gb).cpu().numpy()
    peaks = eng.forward(gb, training=True, noise=torch.zeros(N * K, device=dev),
                        dropout_mask=torch.full((N * 32,), 1.25, device=dev)).cpu().numpy()
    rng = np.random.default_rng(1)
    eng.backward(torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev))
    torch.cuda.synchronize()
    return inf, peaks, {k: v.copy() for k, v in eng.params.grads_dict().items()}


def _batch(n_graphs, n_atoms, K):
    b = small_batch(n_graphs=n_graphs, n_atoms=n_atoms, seed=11)
    if K < 16:
        for k in ("nlist", "edges"):
            b[k] = np.ascontiguousarray(b[k][:, :K])
    return b


@pytest.mark.parametrize("n_graphs,n_atoms,K,E", [(4, 70, 16, 3), (37, 256, 16, 3), (3, 1000, 16, 3), (1, 31, 16, 3), (2, 5, 16, 3),
                                                  (5, 100, 8, 3), (6, 90, 16, 2), (6, 90, 16, 1)])
def test_sixteen_wave_kernels_against_the_eight_wave_ones(gpu_device, monkeypatch, n_graphs, n_atoms, K, E):
    b = _batch(n_graphs, n_atoms, K)
    monkeypatch.setenv("NG_MP_W16", "0")
    inf8, peaks8, g8 = _run(gpu_device, b, E)
    monkeypatch.setenv("NG_MP_W16", "1")
    inf16, peaks16, g16 = _run(gpu_device, b, E)
    scale = max(1.0, np.abs(inf8).max())
    assert np.abs(inf16 - inf8).max() <= 2e-6 * scale
    assert np.abs(peaks16 - peaks8).max() <= 2e-6 * scale
    for k, v in g8.items():
        assert np.abs(g16[k] - v).max() <= 5e-6 * (np.abs(v).max() + 1e-30), k


def test_edge_backward_is_bit_identical_given_the_same_forward(gpu_device, monkeypatch):
    """with the forward fixed (the sixteen-wave one both times) the two edge-side kernels must give every gradient bit for bit"""
    import ctypes as C
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    rng = np.random.default_rng(5)
    N, K, E, F = 1000, 16, 3, 64
    h = rng.standard_normal((N, F)).astype(np.float32)
    nl = np.clip(np.arange(N)[:, None] + rng.integers(-60, 60, (N, K)), 0, N - 1).astype(np.int32)
    e = rng.random((N, K, E)).astype(np.float32)
    inv = (1.0 / K) * np.ones(N, np.float32)
    w = (rng.standard_normal((F, F, E)) * 0.1).astype(np.float32)
    S = rng.random((N, F)).astype(np.float32)
    dH = rng.standard_normal((N, F)).astype(np.float32)
    from nmrgnn_amd.graph import GraphBatch
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NG_MP_W16", mode)
        gb = GraphBatch(np.eye(10, dtype=np.float32)[rng.integers(0, 10, N)], nl.astype(np.int64), e[:, :, 0], inv, device=gpu_device)
        csc_ptr, csc_edge = gb.csc()
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
        th, tn, te, ti, tw, tS, tdH = t(h), t(nl), t(e), t(inv), t(w), t(S), t(dH)
        dh_in = torch.empty(N, F, device=gpu_device)
        de = torch.empty(N * K, E, device=gpu_device)
        dw = torch.empty_like(tw)
        ctx = _lib.get_context(0)
        st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
        ctx.check(ctx.lib.ng_mp_layer_bwd_rec(ctx.handle, st, N, K, F, E, 1, ptr(th), ptr(tn), ptr(te), ptr(ti), ptr(tw), None, ptr(tS),
                                              ptr(csc_ptr), ptr(csc_edge), ptr(tdH), ptr(dh_in), ptr(de), 0, ptr(dw), None), "bwd")
        torch.cuda.synchronize()
        out[mode] = (de.cpu().numpy(), dh_in.cpu().numpy(), dw.cpu().numpy())
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)
