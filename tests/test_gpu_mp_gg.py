"""Gather-GEMM form of the default-width MPLayer (csrc/mp_gw.cuh: mp_gw_kernel, round 5 — 256-row tiles, the gathered
operand formed in registers from an LDS window; the neighbour aggregate never reaches HBM, the backward's node-side pull
gathers dP rows the same way) against a float64 statement of nmrgnn/layers.py:26-46 + model.py:165-167 and against the
aggregate -> HBM -> GEMM kernels it would replace.  (Round 4's producer / consumer kernel left the library in round 6; shapes
the window form does not take — E < 3, padded lists longer than 16 — keep the two-kernel path.)  NG_MP_GG=1 routes every eligible call through the gather-GEMM; by default only the forward of a call that does
not keep the aggregate (inference) on a batch of small graphs takes it."""
import ctypes as C

import numpy as np
import pytest

from helpers import hp_to_oracle, make_hp, randomize_biases, rel_err

pytestmark = pytest.mark.gpu
F = 256


def _ctx():
    from nmrgnn_amd import _lib
    return _lib.get_context(0)


def _st(dev):
    import torch
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def make_lists(rng, N, K, E, p_pad=0.15, span=300):
    """padded lists with index-local neighbours, padded slots (0, 0.0) and compute-side list (padded -> self)"""
    i = np.arange(N)[:, None]
    nl = np.clip(i + rng.integers(-span, span + 1, (N, K)), 0, N - 1).astype(np.int32)
    real = rng.random((N, K)) > p_pad
    e = (rng.standard_normal((N, K, E)) * 0.5 * real[..., None]).astype(np.float32)
    nlc = np.where(real, nl, i).astype(np.int32)
    inv = (1.0 / np.maximum(real.sum(1), 1)).astype(np.float32)
    return nlc, e, inv


def ref_fwd(h, nl, e, inv, w, act=True):
    A = np.einsum('ijn,ijl->inl', e.astype(np.float64), h.astype(np.float64)[nl])          # [N, E, F]
    P = inv[:, None].astype(np.float64) * np.einsum('inl,lmn->im', A, w.astype(np.float64))
    S = softplus(P) if act else P
    return S + h, S


def gpu_fwd(dev, h, nl, e, inv, w, act=1, csr=None, span=0):
    """span: the largest graph of the batch as ng_ctx_set_graph_span announces it (0 = unknown: the context is shared with the
    other test modules, whose engines leave their last batch's value behind)"""
    import torch
    from nmrgnn_amd._lib import ptr
    t = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    N, K = nl.shape if csr is None else (h.shape[0], 0)
    E = w.shape[2]
    th, tw, tinv = t(h), t(w), t(inv)
    out = torch.full((N, F), 7.0, device=dev)
    S = torch.full((N, F), 7.0, device=dev)
    ctx = _ctx()
    ctx.check(ctx.lib.ng_ctx_set_graph_span(ctx.handle, span), "ng_ctx_set_graph_span")
    if csr is None:
        tn, te = t(nl, np.int32), t(e.reshape(-1, E))
        ctx.check(ctx.lib.ng_mp_layer_fwd(ctx.handle, _st(dev), N, K, F, E, act, 1, ptr(th), ptr(tn), ptr(te), ptr(tinv),
                                          ptr(tw), ptr(out), None, ptr(S)), "ng_mp_layer_fwd")
    else:
        rp, col, ev = csr
        trp, tcol, tev = t(rp, np.int32), t(col, np.int32), t(ev)
        ctx.check(ctx.lib.ng_mp_layer_fwd_csr(ctx.handle, _st(dev), N, len(col), F, E, act, 1, ptr(th), ptr(trp), ptr(tcol),
                                              ptr(tev), ptr(tinv), ptr(tw), ptr(out), None, ptr(S)), "ng_mp_layer_fwd_csr")
    torch.cuda.synchronize()
    return out.cpu().numpy().astype(np.float64), S.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("N,K,E", [(700, 16, 3), (128, 16, 3), (1000, 12, 2), (333, 8, 1), (9000, 16, 3)])
def test_forward_matches_float64_and_the_old_path(gpu_device, monkeypatch, N, K, E):
    rng = np.random.default_rng(N + E)
    h = rng.standard_normal((N, F)).astype(np.float32)
    w = (rng.standard_normal((F, F, E)) * 0.05).astype(np.float32)
    nl, e, inv = make_lists(rng, N, K, E)
    ref, refS = ref_fwd(h, nl, e, inv, w)
    monkeypatch.setenv("NG_MP_GG_MIN_ROWS", "1")
    monkeypatch.setenv("NG_MP_GG", "1")
    y1, s1 = gpu_fwd(gpu_device, h, nl, e, inv, w)
    monkeypatch.delenv("NG_MP_GG")
    y0, s0 = gpu_fwd(gpu_device, h, nl, e, inv, w)
    if E == 3:
        assert not np.array_equal(y0, y1)                # the switch selected other kernels (E < 3: the window form does not take the call)
    scale = np.abs(ref).max()
    assert np.abs(y1 - ref).max() < 3e-6 * scale and np.abs(s1 - refS).max() < 3e-6 * scale
    assert np.abs(y1 - ref).max() <= 2.0 * np.abs(y0 - ref).max() + 1e-7 * scale


def test_forward_over_csr_lists_with_rows_longer_than_the_staged_part(gpu_device, monkeypatch):
    """variable degree up to 60: a 128-row tile then holds more entries than the kernel stages in LDS"""
    rng = np.random.default_rng(3)
    N, E = 1500, 3
    deg = rng.integers(0, 61, N)
    deg[:128] = 60                                        # first tile: 7680 entries > 2304 staged
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = np.concatenate([np.sort(rng.integers(0, N, d)) for d in deg]).astype(np.int32)
    ev = (rng.standard_normal((len(col), E)) * 0.3).astype(np.float32)
    h = rng.standard_normal((N, F)).astype(np.float32)
    w = (rng.standard_normal((F, F, E)) * 0.05).astype(np.float32)
    inv = (1.0 / np.maximum(deg, 1)).astype(np.float32)
    A = np.zeros((N, E, F))
    rows = np.repeat(np.arange(N), deg)
    np.add.at(A, rows, ev.astype(np.float64)[:, :, None] * h.astype(np.float64)[col][:, None, :])
    P = inv[:, None] * np.einsum('inl,lmn->im', A, w.astype(np.float64))
    ref = softplus(P) + h
    monkeypatch.setenv("NG_MP_GG_MIN_ROWS", "1")
    monkeypatch.setenv("NG_MP_GG", "1")
    y1, _ = gpu_fwd(gpu_device, h, None, None, inv, w, csr=(rp, col, ev))
    assert np.abs(y1 - ref).max() < 3e-6 * np.abs(ref).max()


def test_operands_beyond_the_fp16_range_are_repaired_in_fp32(gpu_device, monkeypatch):
    rng = np.random.default_rng(5)
    N, K, E = 600, 16, 3
    h = (rng.standard_normal((N, F)) * 1e5).astype(np.float32)           # |A| far beyond 65504
    w = (rng.standard_normal((F, F, E)) * 0.05).astype(np.float32)
    nl, e, inv = make_lists(rng, N, K, E)
    ref, _ = ref_fwd(h, nl, e, inv, w, act=False)
    monkeypatch.setenv("NG_MP_GG_MIN_ROWS", "1")
    monkeypatch.setenv("NG_MP_GG", "1")
    y1, _ = gpu_fwd(gpu_device, h, nl, e, inv, w, act=0)
    assert np.all(np.isfinite(y1))
    assert np.abs(y1 - ref).max() < 5e-6 * np.abs(ref).max()


def _layer_bwd(dev, N, K, E, h, nl, e, inv, w, S, dH, accum_de=None):
    import torch
    from nmrgnn_amd._lib import ptr
    from nmrgnn_amd.graph import GraphBatch
    t = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    atoms = np.zeros((N, 10), np.float32)
    atoms[:, 2] = 1
    gb = GraphBatch(atoms, nl, (np.abs(e).sum(-1) > 0).astype(np.float32), inv, device=dev)
    csc_ptr, csc_edge = gb.csc()
    th, tn, te, tinv, tw, tS, tdH = t(h), t(nl, np.int32), t(e.reshape(-1, E)), t(inv), t(w), t(S), t(dH)
    dh = torch.full((N, F), 7.0, device=dev)
    de = torch.full((N * K, E), 7.0, device=dev)
    dw = torch.full((F, F, E), 7.0, device=dev)
    ctx = _ctx()
    ctx.lib.ng_ctx_set_graph_span(ctx.handle, 0)
    rec = torch.empty(N * K, 4, device=dev)
    ctx.check(ctx.lib.ng_mp_edge_records(ctx.handle, _st(dev), N, K, E, ptr(csc_ptr), ptr(csc_edge), ptr(te), ptr(rec)), "rec")
    ctx.check(ctx.lib.ng_mp_layer_bwd_rec(ctx.handle, _st(dev), N, K, F, E, 1, ptr(th), ptr(tn), ptr(te), ptr(tinv), ptr(tw),
                                          None, ptr(tS), ptr(csc_ptr), ptr(csc_edge), ptr(tdH), ptr(dh), ptr(de), 0, ptr(dw),
                                          ptr(rec)), "ng_mp_layer_bwd_rec")
    torch.cuda.synchronize()
    return dh.cpu().numpy().astype(np.float64), de.cpu().numpy().astype(np.float64), dw.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("N,K,E", [(900, 16, 3), (300, 16, 2)])
def test_backward_pull_matches_float64_and_scales_exactly(gpu_device, monkeypatch, N, K, E):
    rng = np.random.default_rng(N)
    h = rng.standard_normal((N, F)).astype(np.float32)
    w = (rng.standard_normal((F, F, E)) * 0.05).astype(np.float32)
    nl, e, inv = make_lists(rng, N, K, E)
    _, S = ref_fwd(h, nl, e, inv, w)
    dH = rng.standard_normal((N, F)).astype(np.float32)
    # float64 reference of the layer's backward (layers.py:26-46 differentiated)
    h64, e64, w64 = h.astype(np.float64), e.astype(np.float64), w.astype(np.float64)
    dP = dH * (1.0 - np.exp(-S)) * inv[:, None]
    dA = np.einsum('im,lmn->inl', dP, w64)
    A = np.einsum('ijn,ijl->inl', e64, h64[nl])
    ref_dw = np.einsum('inl,im->lmn', A, dP)
    ref_de = np.einsum('inl,ijl->ijn', dA, h64[nl]).reshape(N * K, E)
    ref_dh = dH.astype(np.float64).copy()
    np.add.at(ref_dh, nl.reshape(-1), np.einsum('ijn,inl->ijl', e64, dA).reshape(N * K, F))
    monkeypatch.setenv("NG_MP_GG_MIN_ROWS", "1")
    monkeypatch.setenv("NG_MP_GG", "1")
    dh1, de1, dw1 = _layer_bwd(gpu_device, N, K, E, h, nl, e, inv, w, S.astype(np.float32), dH)
    monkeypatch.delenv("NG_MP_GG")
    dh0, de0, dw0 = _layer_bwd(gpu_device, N, K, E, h, nl, e, inv, w, S.astype(np.float32), dH)
    monkeypatch.setenv("NG_MP_GG", "1")
    if E == 3:
        assert not np.array_equal(dh0, dh1)
    for got, old, ref in ((dh1, dh0, ref_dh), (de1, de0, ref_de), (dw1, dw0, ref_dw)):
        sc = np.abs(ref).max()
        assert np.abs(got - ref).max() < 1e-5 * sc
        assert np.abs(got - ref).max() <= 3.0 * np.abs(old - ref).max() + 1e-6 * sc
    # homogeneity: the pull runs on S * dP with a power of two S chosen per call -> dH * 2^k gives the bits of dh * 2^k
    for k in (-40, 24):
        dhk, _, _ = _layer_bwd(gpu_device, N, K, E, h, nl, e, inv, w, S.astype(np.float32), dH * np.float32(2.0 ** k))
        assert np.array_equal(dhk, dh1 * 2.0 ** k)


def test_model_at_the_default_width_through_the_gather_gemm(gpu_device, monkeypatch):
    """whole model (F = 256): forward + every gradient against the float64 oracle with every MPLayer on the gather-GEMM"""
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    from nmrgnn_amd import synth
    from oracle import nmrgnn_oracle as O
    monkeypatch.setenv("NG_MP_GG_MIN_ROWS", "1")
    monkeypatch.setenv("NG_MP_GG", "1")
    hp = make_hp(atom_feature_size=F)
    b = synth.make_batch(3, 150, 16, 10, 0.1, seed=4)
    eng = Engine(hp, 10, device=gpu_device, seed=2)
    sd = randomize_biases(eng)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    N, K = b["edges"].shape
    xi = eng.randn(N * K, seed=1)
    mask = eng.dropout_mask(N * (F // 2), seed=2)
    peaks = eng.forward(gb, training=True, noise=xi, dropout_mask=mask)
    dp = torch.from_numpy(np.random.default_rng(0).standard_normal(N).astype(np.float32)).to(gpu_device)
    eng.backward(dp)
    torch.cuda.synchronize()
    ref, grads = O.gnn_forward_backward((b["atoms"], b["nlist"], b["edges"], b["inv_degree"]), sd, hp_to_oracle(hp),
                                        dp.cpu().numpy().astype(np.float64), training=True,
                                        noise=xi.cpu().numpy().reshape(N, K), dropout_mask=(mask.cpu().numpy().reshape(N, F // 2) > 0))
    assert np.abs(peaks.cpu().numpy() - ref).max() < 1e-4
    g = eng.params.grads_dict()
    for k, v in grads.items():
        assert rel_err(g[k], v) < 2e-4, k


def test_full_size_default_width_step_against_the_oracle_on_a_sample(gpu_device):
    """BASELINE configs[2]'s batch (512 x 256 atoms) at the reference's DEFAULT width: the kernels run at the 131k-row shape
    the bench times; peaks of 16 sampled graphs and every gradient of a loss that weighs only those graphs against the
    float64 oracle on the 16-graph sub-batch (graphs are independent: nmrgnn/model.py is per atom)."""
    import torch
    from nmrgnn_amd import synth
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    from helpers import oracle_batch_forward_backward
    hp = make_hp(atom_feature_size=F)
    b = synth.make_batch(512, 256, 16, 10, 0.05, seed=42)
    eng = Engine(hp, 10, device=gpu_device, seed=1234)
    sd = randomize_biases(eng, scale=0.05)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    N, K = b["edges"].shape
    xi = eng.randn(N * K, seed=7)
    mask = eng.dropout_mask(N * (F // 2), seed=8)
    sample = np.sort(np.random.default_rng(1).choice(512, 16, replace=False))
    rows = np.concatenate([np.arange(g * 256, (g + 1) * 256) for g in sample])
    dpe = np.zeros(N, np.float32)
    dpe[rows] = np.random.default_rng(2).standard_normal(len(rows)).astype(np.float32) / 512
    peaks = eng.forward(gb, training=True, noise=xi, dropout_mask=mask)
    eng.backward(torch.from_numpy(dpe).to(gpu_device))
    torch.cuda.synchronize()
    pk = peaks.cpu().numpy()
    grads = eng.params.grads_dict()
    # the 16-graph sub-batch with local indices
    sub = dict(atoms=b["atoms"][rows], edges=b["edges"][rows], inv_degree=b["inv_degree"][rows],
               nlist=np.concatenate([b["nlist"][g * 256:(g + 1) * 256] - g * 256 + i * 256 for i, g in enumerate(sample)]),
               graph_ptr=np.arange(17) * 256)
    xi_h = xi.cpu().numpy().reshape(N, K).astype(np.float64)[rows]
    mk_h = (mask.cpu().numpy().reshape(N, F // 2) > 0).astype(np.float64)[rows]
    ref_pk, ref_g = oracle_batch_forward_backward(sub, sd, hp_to_oracle(hp), dpe[rows].astype(np.float64), xi=xi_h, mask=mk_h,
                                                  workers=8)
    assert np.max(np.abs(pk[rows] - ref_pk)) < 1e-4
    errs = {k: rel_err(grads[k], g) for k, g in ref_g.items()}
    print("full-size F=256 gradient rel. errors:", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < 2e-4, errs


def _graph_local_lists(rng, N, K, E, G=256, p_pad=0.15):
    """padded lists whose neighbours lie in the row's own block of G rows (a batch of G-atom graphs)"""
    i = np.arange(N)[:, None]
    g0 = (i // G) * G
    nl = np.minimum(g0 + rng.integers(0, G, (N, K)), N - 1).astype(np.int32)
    real = rng.random((N, K)) > p_pad
    e = (rng.standard_normal((N, K, E)) * 0.5 * real[..., None]).astype(np.float32)
    nlc = np.where(real, nl, i).astype(np.int32)
    inv = (1.0 / np.maximum(real.sum(1), 1)).astype(np.float32)
    return nlc, e, inv


@pytest.mark.parametrize("N", [8192 + 40, 8192 + 64, 9000])
def test_window_form_is_the_default_for_inference_on_molecule_batches(gpu_device, monkeypatch, N):
    """No switch set: with the batch's largest graph announced (ng_ctx_set_graph_span) and no aggregate kept, the forward runs
    mp_gw_kernel.  The last tile has 40 / 64 / 232 rows: one of its four waves owns every valid row, the others race ahead
    (the shapes that exposed a missing barrier between the first k-step's gather and the next window request)."""
    rng = np.random.default_rng(N)
    K, E = 16, 3
    h = rng.standard_normal((N, F)).astype(np.float32)
    w = (rng.standard_normal((F, F, E)) * 0.05).astype(np.float32)
    nl, e, inv = _graph_local_lists(rng, N, K, E)
    ref, refS = ref_fwd(h, nl, e, inv, w)
    monkeypatch.setenv("NG_MP_GG_MIN_ROWS", "1")
    y1, s1 = gpu_fwd(gpu_device, h, nl, e, inv, w, span=256)
    monkeypatch.setenv("NG_MP_GG_MIN_ROWS", "1000000000")      # no call is large enough: aggregate -> HBM -> GEMM
    y0, s0 = gpu_fwd(gpu_device, h, nl, e, inv, w, span=256)
    _ctx().lib.ng_ctx_set_graph_span(_ctx().handle, 0)
    assert not np.array_equal(y0, y1)                    # the default selected the window kernel
    scale = np.abs(ref).max()
    assert np.abs(y1 - ref).max() < 3e-6 * scale and np.abs(s1 - refS).max() < 3e-6 * scale
    assert np.abs(y1 - ref).max() <= 2.0 * np.abs(y0 - ref).max() + 1e-7 * scale


def test_window_form_with_lists_longer_than_its_staging_area(gpu_device, monkeypatch):
    """CSR rows of up to 60 graph-local entries: the first tile's tails (entries 16..) exceed the 480 staged records, the rest
    is read from memory inside the window form"""
    rng = np.random.default_rng(11)
    N, E, G = 1024, 3, 256
    deg = rng.integers(0, 25, N)
    deg[:200] = 60
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = np.concatenate([np.sort((r // G) * G + rng.integers(0, G, d)) for r, d in enumerate(deg)]).astype(np.int32)
    ev = (rng.standard_normal((len(col), E)) * 0.3).astype(np.float32)
    h = rng.standard_normal((N, F)).astype(np.float32)
    w = (rng.standard_normal((F, F, E)) * 0.05).astype(np.float32)
    inv = (1.0 / np.maximum(deg, 1)).astype(np.float32)
    A = np.zeros((N, E, F))
    rows = np.repeat(np.arange(N), deg)
    np.add.at(A, rows, ev.astype(np.float64)[:, :, None] * h.astype(np.float64)[col][:, None, :])
    P = inv[:, None] * np.einsum('inl,lmn->im', A, w.astype(np.float64))
    ref = softplus(P) + h
    monkeypatch.setenv("NG_MP_GG_MIN_ROWS", "1")
    monkeypatch.setenv("NG_MP_GG", "1")
    y1, _ = gpu_fwd(gpu_device, h, None, None, inv, w, csr=(rp, col, ev))
    assert np.abs(y1 - ref).max() < 3e-6 * np.abs(ref).max()


def test_window_and_memory_paths_of_the_window_form_give_the_same_bits(gpu_device, monkeypatch):
    """A tile whose sources do not fit the LDS window reads them from memory with the same code: same entries in the same
    order, same arithmetic.  NG_MP_GW=nowin sends every tile down that path: forward and pull must not move by a bit; and a
    second run of either is bit-identical (no atomics, no order left to the scheduler)."""
    rng = np.random.default_rng(77)
    N, K, E = 2048 + 100, 16, 3
    h = rng.standard_normal((N, F)).astype(np.float32)
    w = (rng.standard_normal((F, F, E)) * 0.05).astype(np.float32)
    nl, e, inv = _graph_local_lists(rng, N, K, E)
    monkeypatch.setenv("NG_MP_GG_MIN_ROWS", "1")
    monkeypatch.setenv("NG_MP_GG", "1")
    y_win, s_win = gpu_fwd(gpu_device, h, nl, e, inv, w)
    y_again, _ = gpu_fwd(gpu_device, h, nl, e, inv, w)
    dH = rng.standard_normal((N, F)).astype(np.float32)
    S = s_win.astype(np.float32)
    g_win = _layer_bwd(gpu_device, N, K, E, h, nl, e, inv, w, S, dH)
    monkeypatch.setenv("NG_MP_GW", "nowin")
    y_mem, s_mem = gpu_fwd(gpu_device, h, nl, e, inv, w)
    g_mem = _layer_bwd(gpu_device, N, K, E, h, nl, e, inv, w, S, dH)
    assert np.array_equal(y_win, y_again)
    assert np.array_equal(y_win, y_mem) and np.array_equal(s_win, s_mem)
    for a, b in zip(g_win, g_mem):
        assert np.array_equal(a, b)
