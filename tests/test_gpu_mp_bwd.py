"""ng_mp_layer_bwd_rec on the F = 64 paths against a float64 numpy statement of SURVEY App. B:
local windows, hub atoms (in-degree far above 16: several gather rounds, record staging overflow),
unrestricted lists (global-gather branch), ragged N, K in {8, 16}, E in {1, 2, 3}."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def ref_bwd(h, nl, e, inv, w, dH, act):
    N, K = nl.shape
    A = np.einsum("ijn,ijl->inl", e, h[nl])
    P = inv[:, None] * np.einsum("inl,lmn->im", A, w)
    if act == 1:
        sig = 1.0 / (1.0 + np.exp(-P))
        S = np.log1p(np.exp(-np.abs(P))) + np.maximum(P, 0)
    else:
        sig = np.ones_like(P)
        S = P
    dP = dH * sig * inv[:, None]
    dw = np.einsum("inl,im->lmn", A, dP)
    dA = np.einsum("im,lmn->inl", dP, w)
    de = np.einsum("inl,ijl->ijn", dA, h[nl])
    dh = dH.copy()
    np.add.at(dh, nl.reshape(-1), np.einsum("ijn,inl->ijl", e, dA).reshape(N * K, -1))
    return S, de, dh, dw


def make_case(kind, N, K, E, rng):
    if kind == "local":
        nl = np.clip(np.arange(N)[:, None] + rng.integers(-60, 60, (N, K)), 0, N - 1)
    elif kind == "hub":            # everybody points into the first tile: in-degree ~ N*K/32 per target
        nl = rng.integers(0, 32, (N, K))
        nl[:, K // 2:] = np.clip(np.arange(N)[:, None] + rng.integers(-40, 40, (N, K - K // 2)), 0, N - 1)
    else:
        nl = rng.integers(0, N, (N, K))
    e = rng.standard_normal((N, K, E))
    pad = rng.random((N, K)) < 0.1
    e[pad] = 0.0
    return nl.astype(np.int32), e


@pytest.mark.parametrize("kind,N,K,E,act", [("local", 1000, 16, 3, 1), ("hub", 1500, 16, 3, 1), ("wide", 700, 16, 3, 1),
                                            ("local", 333, 8, 2, 0), ("hub", 300, 16, 1, 1), ("local", 31, 16, 3, 1)])
@pytest.mark.parametrize("path", ["default", "layered"])
def test_mp_layer_bwd_vs_numpy(gpu_device, monkeypatch, path, kind, N, K, E, act):
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    from nmrgnn_amd.graph import GraphBatch
    if path == "layered":
        monkeypatch.setenv("NG_MP_PATH", "layered")
    rng = np.random.default_rng(N + 7 * K + E)
    F = 64
    nl, e = make_case(kind, N, K, E, rng)
    h = rng.standard_normal((N, F)) * 0.5
    inv = rng.random(N)
    w = rng.standard_normal((F, F, E)) * 0.1
    dH = rng.standard_normal((N, F))
    S, de, dh, dw = ref_bwd(h, nl, e, inv, w, dH, act)

    dev = gpu_device
    t = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    # incoming-edge lists exactly as the engine builds them (edges > 0 slots only)
    edges = (np.abs(e).sum(-1) > 0).astype(np.float32)
    gb = GraphBatch(np.eye(10, dtype=np.float32)[rng.integers(0, 10, N)], nl, edges, inv, device=dev)
    csc_ptr, csc_edge = gb.csc()
    th, te, tinv, tw, tdH, tS = t(h), t(e), t(inv), t(w), t(dH), t(S)
    tdh = torch.empty(N, F, device=dev)
    tde = torch.full((N, K, E), 0.25, device=dev)
    tdw = torch.empty(F, F, E, device=dev)
    rec = torch.empty(N * K, 4, device=dev)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.ng_mp_edge_records(ctx.handle, st, N, K, E, ptr(csc_ptr), ptr(csc_edge), ptr(te), ptr(rec)), "rec")
    ctx.check(ctx.lib.ng_mp_layer_bwd_rec(ctx.handle, st, N, K, F, E, act, ptr(th), ptr(gb.nlist_c), ptr(te), ptr(tinv),
                                          ptr(tw), None, ptr(tS), ptr(csc_ptr), ptr(csc_edge), ptr(tdH), ptr(tdh),
                                          ptr(tde), 1, ptr(tdw), ptr(rec)), "bwd")
    scale = lambda a: max(1.0, np.abs(a).max())
    assert np.abs(tdh.cpu().numpy() - dh).max() < 2e-4 * scale(dh)
    assert np.abs(tdw.cpu().numpy() - dw).max() < 2e-4 * scale(dw)
    live = np.abs(e).sum(-1) > 0                                   # masked slots: de is unspecified
    got = tde.cpu().numpy() - 0.25                                 # accumulate = 1
    assert np.abs((got - de)[live]).max() < 2e-4 * scale(de)
    # without supplied records the call rebuilds them
    tdh2 = torch.empty_like(tdh); tdw2 = torch.empty_like(tdw); tde2 = torch.zeros(N, K, E, device=dev)
    ctx.check(ctx.lib.ng_mp_layer_bwd(ctx.handle, st, N, K, F, E, act, ptr(th), ptr(gb.nlist_c), ptr(te), ptr(tinv),
                                      ptr(tw), None, ptr(tS), ptr(csc_ptr), ptr(csc_edge), ptr(tdH), ptr(tdh2),
                                      ptr(tde2), 0, ptr(tdw2)), "bwd2")
    assert torch.equal(tdh2, tdh) and torch.equal(tdw2, tdw)


@pytest.mark.parametrize("shift", [-50, -20, 12, 35])
def test_mp_layer_bwd_scales_exactly_with_the_upstream_gradient(gpu_device, shift):
    """The window edge kernel forms dA = dP Wp^T on the fp16 pipe with two-piece operands and a power-of-two scale PER ATOM
    ROW taken from the row's max |dP| (mp_win_bwd.hip), so tiny gradients keep their bits.  Property: dH * 2^shift gives
    de, dh and dw times 2^shift bit for bit (fp32 arithmetic is homogeneous under powers of two away from under/overflow)."""
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    from nmrgnn_amd.graph import GraphBatch
    N, K, E, F, act = 1000, 16, 3, 64, 1
    rng = np.random.default_rng(3)
    nl, e = make_case("local", N, K, E, rng)
    h = rng.standard_normal((N, F)) * 0.5
    inv = rng.random(N)
    w = rng.standard_normal((F, F, E)) * 0.1
    dH = rng.standard_normal((N, F)).astype(np.float32)
    dH[5] = 0.0                                             # an all-zero row keeps S = 1
    S, _, _, _ = ref_bwd(h, nl, e, inv, w, dH.astype(np.float64), act)
    dev = gpu_device
    t = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    edges = (np.abs(e).sum(-1) > 0).astype(np.float32)
    gb = GraphBatch(np.eye(10, dtype=np.float32)[rng.integers(0, 10, N)], nl, edges, inv, device=dev)
    csc_ptr, csc_edge = gb.csc()
    th, te, tinv, tw, tS = t(h), t(e), t(inv), t(w), t(S)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def run(dh_up):
        tdH = t(dh_up)
        tdh = torch.empty(N, F, device=dev); tdw = torch.empty(F, F, E, device=dev); tde = torch.zeros(N, K, E, device=dev)
        ctx.check(ctx.lib.ng_mp_layer_bwd(ctx.handle, st, N, K, F, E, act, ptr(th), ptr(gb.nlist_c), ptr(te), ptr(tinv),
                                          ptr(tw), None, ptr(tS), ptr(csc_ptr), ptr(csc_edge), ptr(tdH), ptr(tdh),
                                          ptr(tde), 0, ptr(tdw)), "bwd")
        torch.cuda.synchronize()
        return [x.cpu().numpy().astype(np.float64) for x in (tdh, tdw, tde)]

    base = run(dH)
    moved = run((dH.astype(np.float64) * 2.0 ** shift).astype(np.float32))
    live = np.abs(e).sum(-1) > 0
    for k, (a, b) in enumerate(zip(moved, base)):
        if k == 2:
            a, b = a[live], b[live]
        assert np.isfinite(a).all()
        assert np.array_equal(a, b * 2.0 ** shift), k


@pytest.mark.parametrize("graph,K,E", [(256, 16, 3), (200, 8, 2)])
def test_default_width_backward_with_slab_windows(gpu_device, graph, K, E):
    """F = 256 (the reference's default): with the locality hint (ng_ctx_set_graph_span) the neighbour aggregation and the
    edge gradient de = <dA, h[nlist]> keep slab windows of h in LDS (mp_win.hip: agg_win_kernel, egrad_win_kernel).  Both
    forms of ng_mp_layer_bwd against the float64 statement of SURVEY App. B and against each other; graphs of 200 atoms
    straddle the 32-atom tiles (window restaging / global fall-back), the last tile is ragged."""
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    from nmrgnn_amd.graph import GraphBatch
    F, act = 256, 1
    G = 4200 // graph + 1
    N = G * graph - 5
    rng = np.random.default_rng(graph + K)
    base = (np.arange(N) // graph) * graph
    nl = np.minimum(base[:, None] + rng.integers(0, graph, (N, K)), N - 1).astype(np.int32)
    e = rng.standard_normal((N, K, E))
    e[rng.random((N, K)) < 0.1] = 0.0
    h = rng.standard_normal((N, F)) * 0.5
    inv = rng.random(N)
    w = rng.standard_normal((F, F, E)) * 0.05
    dH = rng.standard_normal((N, F))
    S, de, dh, dw = ref_bwd(h, nl, e, inv, w, dH, act)
    dev = gpu_device
    t = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    edges = (np.abs(e).sum(-1) > 0).astype(np.float32)
    gb = GraphBatch(np.eye(10, dtype=np.float32)[rng.integers(0, 10, N)], nl, edges, inv, device=dev)
    csc_ptr, csc_edge = gb.csc()
    th, te, tinv, tw, tdH, tS = t(h), t(e), t(inv), t(w), t(dH), t(S)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    live = np.abs(e).sum(-1) > 0
    got = {}
    for span in (0, graph):
        ctx.check(ctx.lib.ng_ctx_set_graph_span(ctx.handle, span), "span")
        tdh = torch.empty(N, F, device=dev); tdw = torch.empty(F, F, E, device=dev); tde = torch.full((N, K, E), 0.5, device=dev)
        ctx.check(ctx.lib.ng_mp_layer_bwd(ctx.handle, st, N, K, F, E, act, ptr(th), ptr(gb.nlist_c), ptr(te), ptr(tinv),
                                          ptr(tw), None, ptr(tS), ptr(csc_ptr), ptr(csc_edge), ptr(tdH), ptr(tdh),
                                          ptr(tde), 1, ptr(tdw)), "bwd")
        torch.cuda.synchronize()
        got[span] = (tdh.cpu().numpy(), tdw.cpu().numpy(), tde.cpu().numpy() - 0.5)
    ctx.check(ctx.lib.ng_ctx_set_graph_span(ctx.handle, 0), "span")
    scale = lambda a: max(1.0, np.abs(a).max())
    for span, (gdh, gdw, gde) in got.items():
        assert np.abs(gdh - dh).max() < 2e-4 * scale(dh), span
        assert np.abs(gdw - dw).max() < 2e-4 * scale(dw), span
        assert np.abs((gde - de)[live]).max() < 2e-4 * scale(de), span
    # the two forms differ only by the summation order of the window kernels
    assert np.abs((got[0][2] - got[graph][2])[live]).max() < 1e-5 * scale(de)
