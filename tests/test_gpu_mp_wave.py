"""The wave-autonomous forward window kernel (round 6: csrc/mp_wave.hip — default for E = 3 on batches that fill the chip,
NG_MP_WAVE=1 / 0 forces / forbids it) against float64 and against the sixteen-wave kernel it replaces (mp_win16.hip).
Reference: nmrgnn/layers.py:26-46 + the residual of nmrgnn/model.py:165-167.
  same per-atom sums in list order, the k-steps of the contraction in order: agreement with mp_win16.hip to fp32 rounding.
Shapes: several window groups per workgroup, a tail micro-tile, graphs smaller than a micro-tile, K < 16, sources outside the
window (taken from memory: same sums), weights beyond the fp16 piece range (f32-input body), rows that reach 2^15."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_hp, small_batch

pytestmark = pytest.mark.gpu

F = 64


def _case(N, K, spread, seed, trigger="none"):
    rng = np.random.default_rng(seed)
    h = (rng.standard_normal((N, F)) * 0.5).astype(np.float32)
    nl = np.clip(np.arange(N)[:, None] + rng.integers(-spread, spread + 1, (N, K)), 0, N - 1).astype(np.int32)
    e = rng.standard_normal((N, K, 3)).astype(np.float32)
    e[rng.random((N, K)) < 0.1] = 0.0
    inv = (0.05 + rng.random(N)).astype(np.float32)
    w = (rng.standard_normal((F, F, 3)) * 0.1).astype(np.float32)
    if trigger == "weights":
        w[3, 5, 0] = 400.0
        w[40, 63, 2] = -300.0
    elif trigger == "features":
        h[rng.integers(0, N, 7), rng.integers(0, F, 7)] = 1.0e5
    return h, nl, e, inv, w


def _ref(h, nl, e, inv, w, act):
    h64, e64, w64 = h.astype(np.float64), e.astype(np.float64), w.astype(np.float64)
    A = np.einsum("ijn,ijl->inl", e64, h64[nl])
    P = inv.astype(np.float64)[:, None] * np.einsum("inl,lmn->im", A, w64)
    mag = inv.astype(np.float64)[:, None] * np.einsum("inl,lmn->im", np.einsum("ijn,ijl->inl", np.abs(e64), np.abs(h64)[nl]),
                                                      np.abs(w64))
    S = (np.log1p(np.exp(-np.abs(P))) + np.maximum(P, 0)) if act else P
    return S, S + h64, mag


def _fwd(gpu_device, h, nl, e, inv, w, act, save=True):
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    N, K, E = e.shape
    th, tn, te, ti, tw = (torch.from_numpy(x).to(gpu_device) for x in (h, nl, e, inv, w))
    out = torch.full((N, F), 7.0, device=gpu_device)
    S = torch.full((N, F), 7.0, device=gpu_device)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
    ctx.check(ctx.lib.ng_mp_layer_fwd(ctx.handle, st, N, K, F, E, act, 1, ptr(th), ptr(tn), ptr(te), ptr(ti), ptr(tw), ptr(out),
                                      None, ptr(S) if save else None), "mp")
    torch.cuda.synchronize()
    return out.cpu().numpy(), S.cpu().numpy()


@pytest.mark.parametrize("N,K,spread,act", [(1000, 16, 50, 1), (4099, 16, 120, 1), (333, 8, 40, 0), (70000, 16, 100, 1),
                                            (5, 4, 2, 1), (2050, 12, 1500, 1), (300, 16, 300, 0)])
def test_wave_forward_against_float64_and_the_sixteen_wave_kernel(gpu_device, monkeypatch, N, K, spread, act):
    case = _case(N, K, spread, N + K)
    S_ref, out_ref, mag = _ref(*case, act)
    monkeypatch.setenv("NG_MP_WAVE", "0")
    out16, S16 = _fwd(gpu_device, *case, act)
    monkeypatch.setenv("NG_MP_WAVE", "1")
    out, S = _fwd(gpu_device, *case, act)
    tol = 3e-6 * max(1.0, mag.max())
    assert np.isfinite(out).all()
    assert np.abs(out - out_ref).max() < tol
    assert np.abs(S - S_ref).max() < tol
    assert np.abs(out - out16).max() <= 2e-6 * max(1.0, mag.max())
    assert np.abs(S - S16).max() <= 2e-6 * max(1.0, mag.max())
    # without the activation copy the outputs are the same bits
    out_ns, S_ns = _fwd(gpu_device, *case, act, save=False)
    assert np.array_equal(out_ns, out)
    assert (S_ns == 7.0).all()
    # run to run
    out2, S2 = _fwd(gpu_device, *case, act)
    assert np.array_equal(out2, out) and np.array_equal(S2, S)


@pytest.mark.parametrize("trigger", ["weights", "features"])
def test_wave_forward_beyond_the_piece_range(gpu_device, monkeypatch, trigger):
    case = _case(3000, 16, 60, 17, trigger)
    S_ref, out_ref, mag = _ref(*case, 1)
    monkeypatch.setenv("NG_MP_WAVE", "1")
    out, S = _fwd(gpu_device, *case, 1)
    assert np.isfinite(out).all()
    assert np.abs(out - out_ref).max() < 3e-6 * max(1.0, mag.max())
    assert np.abs(S - S_ref).max() < 3e-6 * max(1.0, mag.max())


@pytest.mark.parametrize("n_graphs,n_atoms,K", [(4, 70, 16), (37, 256, 16), (3, 1000, 16), (1, 31, 16), (5, 100, 8)])
def test_whole_engine_with_the_wave_forward(gpu_device, monkeypatch, n_graphs, n_atoms, K):
    """peaks and every gradient through the engine with the wave kernel forced, against the sixteen-wave forward"""
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    b = small_batch(n_graphs=n_graphs, n_atoms=n_atoms, seed=11)
    if K < 16:
        for k in ("nlist", "edges"):
            b[k] = np.ascontiguousarray(b[k][:, :K])

    def run():
        eng = Engine(make_hp(atom_feature_size=64, edge_feature_size=3), 10, device=gpu_device, seed=3)
        gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
        N, KK = gb.edges.shape
        inf = eng.forward(gb).cpu().numpy()
        peaks = eng.forward(gb, training=True, noise=torch.zeros(N * KK, device=gpu_device),
                            dropout_mask=torch.full((N * 32,), 1.25, device=gpu_device)).cpu().numpy()
        rng = np.random.default_rng(1)
        eng.backward(torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(gpu_device))
        torch.cuda.synchronize()
        return inf, peaks, {k: v.copy() for k, v in eng.params.grads_dict().items()}

    monkeypatch.setenv("NG_MP_WAVE", "0")
    inf0, peaks0, g0 = run()
    monkeypatch.setenv("NG_MP_WAVE", "1")
    inf1, peaks1, g1 = run()
    scale = max(1.0, np.abs(inf0).max())
    assert np.abs(inf1 - inf0).max() <= 2e-6 * scale
    assert np.abs(peaks1 - peaks0).max() <= 2e-6 * scale
    for k, v in g0.items():
        assert np.abs(g1[k] - v).max() <= 5e-6 * (np.abs(v).max() + 1e-30), k

