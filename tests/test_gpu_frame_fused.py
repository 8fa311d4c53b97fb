"""Molecule-sized inference kernels (csrc/frame_fused.hip): the FC block + head in one launch against the layered path
(ng_fc_block_fwd + ng_head_fwd) and against a float64 statement of nmrgnn/model.py:191-196,268-273."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F, Lf, NC = 256, 4, 10


def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def _case(N, seed, big_rows=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, F)).astype(np.float32)
    if big_rows:
        rows = rng.choice(N, big_rows, replace=False)
        x[rows, rng.integers(0, F, big_rows)] = 2.0e5            # beyond the fp16 range of a piece
    W = [(rng.standard_normal((F, F)) * 0.06).astype(np.float32) for _ in range(3)] + \
        [(rng.standard_normal((F, F // 2)) * 0.06).astype(np.float32)]
    b = [(rng.standard_normal(F) * 0.1).astype(np.float32) for _ in range(3)] + [(rng.standard_normal(F // 2) * 0.1).astype(np.float32)]
    Wout = (rng.standard_normal((F // 2, NC)) * 0.1).astype(np.float32)
    bout = (rng.standard_normal(NC) * 0.1).astype(np.float32)
    elem = rng.integers(0, NC, N)
    atoms = np.eye(NC, dtype=np.float32)[elem]
    std = np.array([0, 0, 10.6, 50.9, 6.04, 0, 1, 2, 0.5, 1], np.float32)
    avg = np.array([0, 0, 126.0, 118.9, 5.63, 0, 1, -2, 3, 0], np.float32)
    return x, W, b, Wout, bout, atoms, std, avg


def _ref(x, W, b, Wout, bout, atoms, std, avg):
    h = x.astype(np.float64)
    for l in range(3):
        h = softplus(h @ W[l].astype(np.float64) + b[l]) + h
    g = softplus(h @ W[3].astype(np.float64) + b[3])
    full = g @ Wout.astype(np.float64) + bout
    return (atoms * (full * std + avg)).sum(1)


def _run(dev, case, frozen):
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr, ptr_array
    x, W, b, Wout, bout, atoms, std, avg = case
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tx, tW, tb = t(x), [t(w) for w in W], [t(v) for v in b]
    tWo, tbo, ta, ts, tv = t(Wout), t(bout), t(atoms), t(std), t(avg)
    N = x.shape[0]
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    peaks = torch.full((N,), 7.0, device=dev)
    if frozen:
        # a fresh owner per case: the cache is keyed by weight ADDRESSES, and torch hands the addresses of the previous
        # case's (freed) weights to this one
        _run.owner = getattr(_run, "owner", 40000) + 1
        ctx.check(ctx.lib.ng_weights_frozen(ctx.handle, _run.owner), "freeze")
    try:
        for _ in range(2 if frozen else 1):        # second call: images served from the cache
            ctx.check(ctx.lib.ng_fc_head_fwd(ctx.handle, st, N, F, Lf, NC, 1, ptr(tx), ptr_array(tW), ptr_array(tb), ptr(tWo), ptr(tbo),
                                             ptr(ta), ptr(ts), ptr(tv), ptr(peaks)), "ng_fc_head_fwd")
    finally:
        ctx.lib.ng_weights_frozen(ctx.handle, 0)
    # the layered path on the same inputs
    ys = [torch.empty(N, F, device=dev) for _ in range(3)]
    g = torch.empty(N, F // 2, device=dev)
    ctx.check(ctx.lib.ng_fc_block_fwd(ctx.handle, st, N, F, Lf, 1, ptr(tx), ptr_array(tW), ptr_array(tb), ptr_array(ys), ptr(g)), "fc")
    lay = torch.empty(N, device=dev)
    ctx.check(ctx.lib.ng_head_fwd(ctx.handle, st, N, F // 2, NC, ptr(g), None, ptr(tWo), ptr(tbo), ptr(ta), ptr(ts), ptr(tv),
                                  ptr(lay)), "head")
    torch.cuda.synchronize()
    return peaks.cpu().numpy().astype(np.float64), lay.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("N,frozen", [(2770, True), (2770, False), (31, False), (33, True), (8000, True)])
def test_fused_fc_head_equals_float64_and_the_layered_path(gpu_device, N, frozen):
    case = _case(N, seed=N)
    ref = _ref(*case)
    fused, lay = _run(gpu_device, case, frozen)
    std_of = case[6][np.argmax(case[5], 1)].astype(np.float64)
    scale = np.maximum(std_of, 1.0)
    assert np.isfinite(fused).all()
    assert np.max(np.abs(fused - ref) / scale) < 3e-5
    assert np.max(np.abs(fused - lay) / scale) < 3e-5
    assert np.all(fused[std_of == 0] == case[7][np.argmax(case[5], 1)][std_of == 0])     # std = 0: exactly avg


def test_fused_fc_head_repairs_rows_beyond_the_fp16_range(gpu_device):
    case = _case(2770, seed=5, big_rows=40)
    ref = _ref(*case)
    fused, lay = _run(gpu_device, case, True)
    std_of = case[6][np.argmax(case[5], 1)].astype(np.float64)
    scale = np.maximum(std_of, 1.0) * np.maximum(1.0, np.abs(ref) / 100.0)
    assert np.isfinite(fused).all() and np.isfinite(lay).all()
    assert np.max(np.abs(fused - ref) / scale) < 1e-3 * 1.0
    assert np.max(np.abs(fused - ref) / np.maximum(np.abs(ref), 1.0)) < 1e-4


def _mp_case(N, K, E, seed, big_rows=0):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal((N, F)).astype(np.float32)
    if big_rows:
        h[rng.choice(N, big_rows, replace=False), 7] = 3.0e5
    nlist = rng.integers(0, N, (N, K)).astype(np.int32)
    e = (rng.standard_normal((N, K, E)) * 0.3).astype(np.float32)
    e[rng.random((N, K)) < 0.1] = 0.0
    inv = (1.0 / rng.integers(1, K + 1, N)).astype(np.float32)
    w = (rng.standard_normal((F, F, E)) * 0.02).astype(np.float32)
    return h, nlist, e, inv, w


@pytest.mark.parametrize("N,K,E,act,big", [(2770, 16, 3, 1, 0), (100, 16, 3, 1, 0), (33, 5, 2, 3, 0), (2770, 16, 3, 1, 25),
                                           (6000, 32, 1, 2, 0)])
def test_fused_mp_layer_equals_float64_and_the_layered_path(gpu_device, N, K, E, act, big):
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    h, nlist, e, inv, w = _mp_case(N, K, E, N + K, big)
    A = np.einsum("ijn,ijl->inl", e.astype(np.float64), h.astype(np.float64)[nlist])            # [N][E][F]
    pre = inv[:, None].astype(np.float64) * np.einsum("inl,lmn->im", A, w.astype(np.float64))
    actf = {1: softplus, 2: lambda x: np.maximum(x, 0), 3: np.tanh}[act]
    ref = actf(pre) + h
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
    th, tn, te, ti, tw = t(h), t(nlist), t(e), t(inv), t(w)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
    out = torch.full((N, F), 7.0, device=gpu_device)
    ctx.check(ctx.lib.ng_mp_layer_fwd_short(ctx.handle, st, N, K, F, E, act, 1, ptr(th), ptr(tn), ptr(te), ptr(ti), ptr(tw), ptr(out)),
              "ng_mp_layer_fwd_short")
    lay = torch.full((N, F), 7.0, device=gpu_device)
    ctx.check(ctx.lib.ng_mp_layer_fwd(ctx.handle, st, N, K, F, E, act, 1, ptr(th), ptr(tn), ptr(te), ptr(ti), ptr(tw), ptr(lay),
                                      None, None), "ng_mp_layer_fwd")
    torch.cuda.synchronize()
    got, lay = out.cpu().numpy().astype(np.float64), lay.cpu().numpy().astype(np.float64)
    # an output is a 768-term sum: its rounding error scales with the sum of |terms| (large aggregates cancel), not with
    # the result
    mag = np.maximum(inv[:, None] * np.einsum("inl,lmn->im", np.abs(A), np.abs(w.astype(np.float64))) + np.abs(h), 1.0)
    assert np.isfinite(got).all()
    assert np.max(np.abs(got - ref) / mag) < 5e-6        # (repaired rows: a plain fp32 chain over 768 terms)
    assert np.max(np.abs(lay - ref) / mag) < 5e-6
    # in-place use is refused (other atoms still gather the input rows)
    assert ctx.lib.ng_mp_layer_fwd_short(ctx.handle, st, N, K, F, E, act, 1, ptr(th), ptr(tn), ptr(te), ptr(ti), ptr(tw), ptr(th)) != 0


@pytest.mark.parametrize("N,E", [(2770, 3), (77, 2)])
def test_fused_mp_layer_over_csr_lists(gpu_device, N, E):
    """variable degree (0 ... 40 entries per row, one row with 300): the CSR form of the one-launch MPLayer"""
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    rng = np.random.default_rng(N)
    deg = rng.integers(0, 41, N)
    deg[3] = 0
    deg[N // 2] = 300
    row_ptr = np.zeros(N + 1, np.int32)
    row_ptr[1:] = np.cumsum(deg)
    nnz = int(row_ptr[-1])
    col = rng.integers(0, N, nnz).astype(np.int32)
    e = (rng.standard_normal((nnz, E)) * 0.3).astype(np.float32)
    h = rng.standard_normal((N, F)).astype(np.float32)
    inv = (1.0 / np.maximum(deg, 1)).astype(np.float32)
    w = (rng.standard_normal((F, F, E)) * 0.02).astype(np.float32)
    rows = np.repeat(np.arange(N), deg)
    A = np.zeros((N, E, F))
    np.add.at(A, rows, e.astype(np.float64)[:, :, None] * h.astype(np.float64)[col][:, None, :])
    pre = inv[:, None].astype(np.float64) * np.einsum("inl,lmn->im", A, w.astype(np.float64))
    ref = softplus(pre) + h
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
    th, tr, tc, te, ti, tw = t(h), t(row_ptr), t(col), t(e), t(inv), t(w)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
    out = torch.full((N, F), 7.0, device=gpu_device)
    ctx.check(ctx.lib.ng_mp_layer_fwd_short_csr(ctx.handle, st, N, F, E, 1, 1, ptr(th), ptr(tr), ptr(tc), ptr(te), ptr(ti), ptr(tw),
                                                ptr(out)), "ng_mp_layer_fwd_short_csr")
    lay = torch.full((N, F), 7.0, device=gpu_device)
    ctx.check(ctx.lib.ng_mp_layer_fwd_csr(ctx.handle, st, N, nnz, F, E, 1, 1, ptr(th), ptr(tr), ptr(tc), ptr(te), ptr(ti), ptr(tw),
                                          ptr(lay), None, None), "ng_mp_layer_fwd_csr")
    torch.cuda.synchronize()
    mag = np.maximum(inv[:, None] * np.einsum("inl,lmn->im", np.abs(A), np.abs(w.astype(np.float64))) + np.abs(h), 1.0)
    for got in (out.cpu().numpy().astype(np.float64), lay.cpu().numpy().astype(np.float64)):
        assert np.isfinite(got).all()
        assert np.max(np.abs(got - ref) / mag) < 5e-6
