"""The F = 64 window kernels (mp_win.hip, mp_win_bwd.hip) on operands outside the fp16 piece range.

The reference's MPLayer is plain fp32 (nmrgnn/layers.py:39-44): a weight of 400 or a feature of 1e5 is an ordinary number
there.  Feature-side operands of the piece kernels (aggregate rows, the h operand of dw, gradient rows) carry per-row
power-of-two scales and cannot leave the fp16 range; a weight with |2^8 w| >= 65504 is found by the pack launch, which
raises the range guard (ng_internal.h: RangeGuard), and each kernel then runs its fp32-input body.
Property: results equal float64 within the fp32 bound; with out-of-range weights they equal the NG_GEMM_MATH=fp32 bits."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = 64


def _case(trigger, N, K, E, seed):
    rng = np.random.default_rng(seed)
    h = (rng.standard_normal((N, F)) * 0.5).astype(np.float32)
    nl = np.clip(np.arange(N)[:, None] + rng.integers(-50, 50, (N, K)), 0, N - 1).astype(np.int32)
    e = rng.standard_normal((N, K, E)).astype(np.float32)
    e[rng.random((N, K)) < 0.1] = 0.0
    inv = (0.05 + rng.random(N)).astype(np.float32)
    w = (rng.standard_normal((F, F, E)) * 0.1).astype(np.float32)
    if trigger == "weights":
        w[3, 5, 0] = 400.0                  # 2^8 w = 102400: no fp16 piece holds it
        w[40, 63, E - 1] = -300.0
    elif trigger == "features":
        h[rng.integers(0, N, 7), rng.integers(0, F, 7)] = 1.0e5       # the aggregate and the dw operand leave the range
    return h, nl, e, inv, w


def _fwd_ref(h, nl, e, inv, w, act):
    h64, e64, w64 = h.astype(np.float64), e.astype(np.float64), w.astype(np.float64)
    A = np.einsum("ijn,ijl->inl", e64, h64[nl])
    P = inv.astype(np.float64)[:, None] * np.einsum("inl,lmn->im", A, w64)
    mag = inv.astype(np.float64)[:, None] * np.einsum("inl,lmn->im", np.einsum("ijn,ijl->inl", np.abs(e64), np.abs(h64)[nl]),
                                                      np.abs(w64))
    S = (np.log1p(np.exp(-np.abs(P))) + np.maximum(P, 0)) if act else P
    return S, S + h64, mag


def _fwd(gpu_device, h, nl, e, inv, w, act):
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
                                      None, ptr(S)), "mp")
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.float64), S.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("trigger", ["weights", "features", "none"])
@pytest.mark.parametrize("N,K,E,act", [(1000, 16, 3, 1), (333, 8, 2, 0), (4100, 16, 1, 1)])
def test_window_forward_beyond_the_piece_range(gpu_device, monkeypatch, trigger, N, K, E, act):
    case = _case(trigger, N, K, E, N + E)
    S_ref, out_ref, mag = _fwd_ref(*case, act)
    res = {}
    for math in ("f16x2", "fp32"):
        monkeypatch.setenv("NG_GEMM_MATH", math)
        res[math] = _fwd(gpu_device, *case, act)
    for math, (out, S) in res.items():
        assert np.isfinite(out).all(), math
        assert np.abs(out - out_ref).max() < 3e-6 * max(1.0, mag.max()), math
        assert np.abs(S - S_ref).max() < 3e-6 * max(1.0, mag.max()), math
    if trigger == "weights":       # the kernel ran its fp32-input body
        if K <= 16 and os.environ.get("NG_MP_W16", "1") != "0":
            # the sixteen-wave forward (mp_win16.hip) sums the two halves of the contraction in separate waves: the same
            # f32-input products as the NG_GEMM_MATH=fp32 kernel, met in another order
            for i in (0, 1):
                assert np.abs(res["f16x2"][i] - res["fp32"][i]).max() <= 1e-6 * max(1.0, mag.max())
        else:
            np.testing.assert_array_equal(res["f16x2"][0], res["fp32"][0])
            np.testing.assert_array_equal(res["f16x2"][1], res["fp32"][1])


def test_window_forward_keeps_the_range_flag_with_a_frozen_image(gpu_device, monkeypatch):
    """ng_weights_frozen: the piece image is packed once; the out-of-range flag travels with it, so the second call (no pack
    launch) still takes the fp32-input body."""
    from nmrgnn_amd import _lib
    monkeypatch.setenv("NG_GEMM_MATH", "f16x2")
    case = _case("weights", 900, 16, 3, 5)
    _, out_ref, mag = _fwd_ref(*case, 1)
    ctx = _lib.get_context(0)
    ctx.check(ctx.lib.ng_weights_frozen(ctx.handle, 424242), "freeze")
    try:
        import torch
        from nmrgnn_amd._lib import ptr
        h, nl, e, inv, w = case
        N, K, E = e.shape
        th, tn, te, ti, tw = (torch.from_numpy(x).to(gpu_device) for x in (h, nl, e, inv, w))
        st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
        outs = []
        for _ in range(3):
            out = torch.full((N, F), 7.0, device=gpu_device)
            ctx.check(ctx.lib.ng_mp_layer_fwd(ctx.handle, st, N, K, F, E, 1, 1, ptr(th), ptr(tn), ptr(te), ptr(ti), ptr(tw),
                                              ptr(out), None, None), "mp")
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy().astype(np.float64))
    finally:
        ctx.check(ctx.lib.ng_weights_frozen(ctx.handle, 0), "thaw")
    for o in outs:
        assert np.abs(o - out_ref).max() < 3e-6 * mag.max()
        np.testing.assert_array_equal(o, outs[0])


def _bwd(gpu_device, h, nl, e, inv, w, dH, S, act, accumulate):
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    from nmrgnn_amd.graph import GraphBatch
    N, K, E = e.shape
    dev = gpu_device
    t = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    edges = (np.abs(e).sum(-1) > 0).astype(np.float32)
    gb = GraphBatch(np.eye(10, dtype=np.float32)[np.arange(N) % 10], nl, edges, inv, device=dev)
    csc_ptr, csc_edge = gb.csc()
    th, te, tinv, tw, tdH, tS = t(h), t(e), t(inv), t(w), t(dH), t(S)
    tdh = torch.full((N, F), 7.0, device=dev)
    tde = torch.full((N, K, E), 0.25 if accumulate else 7.0, device=dev)
    tdw = torch.full((F, F, E), 7.0, device=dev)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.ng_mp_layer_bwd(ctx.handle, st, N, K, F, E, act, ptr(th), ptr(gb.nlist_c), ptr(te), ptr(tinv), ptr(tw),
                                      None, ptr(tS), ptr(csc_ptr), ptr(csc_edge), ptr(tdH), ptr(tdh), ptr(tde), accumulate,
                                      ptr(tdw)), "bwd")
    torch.cuda.synchronize()
    de = tde.cpu().numpy().astype(np.float64) - (0.25 if accumulate else 0.0)
    return tdh.cpu().numpy().astype(np.float64), de, tdw.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("trigger", ["weights", "features", "none"])
@pytest.mark.parametrize("N,K,E,act,accumulate", [(1000, 16, 3, 1, 1), (333, 8, 2, 0, 0)])
def test_window_backward_beyond_the_piece_range(gpu_device, monkeypatch, trigger, N, K, E, act, accumulate):
    h, nl, e, inv, w = _case(trigger, N, K, E, N + 3 * E)
    rng = np.random.default_rng(9)
    dH = rng.standard_normal((N, F)).astype(np.float32)
    h64, e64, w64, inv64, dH64 = (a.astype(np.float64) for a in (h, e, w, inv, dH))
    A = np.einsum("ijn,ijl->inl", e64, h64[nl])
    P = inv64[:, None] * np.einsum("inl,lmn->im", A, w64)
    if act:
        with np.errstate(over="ignore"):
            sig = 1.0 / (1.0 + np.exp(-P))
        S = np.log1p(np.exp(-np.abs(P))) + np.maximum(P, 0)
    else:
        sig, S = np.ones_like(P), P
    S32 = S.astype(np.float32)
    # the kernels take act' from the SAVED fp32 activation output, 1 - exp(-S)
    dP = dH64 * ((1.0 - np.exp(-S32.astype(np.float64))) if act else 1.0) * inv64[:, None]
    dw_ref = np.einsum("inl,im->lmn", A, dP)
    dA = np.einsum("im,lmn->inl", dP, w64)
    de_ref = np.einsum("inl,ijl->ijn", dA, h64[nl])
    dh_ref = dH64.copy()
    np.add.at(dh_ref, nl.reshape(-1), np.einsum("ijn,inl->ijl", e64, dA).reshape(N * K, -1))
    m_dw = np.einsum("inl,im->lmn", np.abs(A), np.abs(dP)).max()
    m_dA = np.einsum("im,lmn->inl", np.abs(dP), np.abs(w64))
    m_de = np.einsum("inl,ijl->ijn", m_dA, np.abs(h64)[nl]).max()
    m_dh = np.abs(dH64).copy()
    np.add.at(m_dh, nl.reshape(-1), np.einsum("ijn,inl->ijl", np.abs(e64), m_dA).reshape(N * K, -1))
    res = {}
    for math in ("f16x2", "fp32"):
        monkeypatch.setenv("NG_GEMM_MATH", math)
        res[math] = _bwd(gpu_device, h, nl, e, inv, w, dH, S32, act, accumulate)
    live = np.abs(e).sum(-1) > 0
    for math, (dh, de, dw) in res.items():
        for got in (dh, de[live], dw):
            assert np.isfinite(got).all(), math
        assert np.abs(dh - dh_ref).max() < 5e-6 * max(1.0, m_dh.max()), math
        assert np.abs(dw - dw_ref).max() < 5e-6 * max(1.0, m_dw), math
        assert np.abs((de - de_ref)[live]).max() < 5e-6 * max(1.0, m_de) + (1e-6 if accumulate else 0.0), math
    if trigger == "weights":       # both kernels ran their fp32-input bodies
        np.testing.assert_array_equal(res["f16x2"][1][live], res["fp32"][1][live])
        np.testing.assert_array_equal(res["f16x2"][0], res["fp32"][0])
        np.testing.assert_array_equal(res["f16x2"][2], res["fp32"][2])
