"""ng_fc_block_fwd / ng_fc_block_bwd (FCBlock of nmrgnn/model.py:179-196, all layers in one call) against a
float64 numpy statement: fused kernels (F = 64, L = 2..6), the per-layer fallback (F = 32), ragged N."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def ref_fc(x, Ws, bs, act):
    f = softplus if act == 1 else (lambda v: v)
    xs = [x]
    for W, b in zip(Ws[:-1], bs[:-1]):
        xs.append(f(xs[-1] @ W + b) + xs[-1])
    return xs, f(xs[-1] @ Ws[-1] + bs[-1])


def ref_fc_bwd(xs, g, Ws, dg, act):
    dact = (lambda s: 1 - np.exp(-s)) if act == 1 else (lambda s: np.ones_like(s))
    L = len(Ws)
    dWs, dbs = [None] * L, [None] * L
    dP = dg * dact(g)
    dWs[L - 1], dbs[L - 1] = xs[L - 1].T @ dP, dP.sum(0)
    d = dP @ Ws[L - 1].T
    for l in range(L - 2, -1, -1):
        s = xs[l + 1] - xs[l]
        dP = d * dact(s)
        dWs[l], dbs[l] = xs[l].T @ dP, dP.sum(0)
        d = d + dP @ Ws[l].T
    return d, dWs, dbs


@pytest.mark.parametrize("N,F,L,act", [(1000, 64, 4, 1), (64, 64, 2, 1), (777, 64, 3, 0), (2049, 64, 6, 1),
                                       (1, 64, 4, 1), (500, 32, 4, 1)])
def test_fc_block_vs_numpy(gpu_device, N, F, L, act):
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr, ptr_array
    rng = np.random.default_rng(N + L)
    Fh = F // 2
    x = rng.standard_normal((N, F))
    Ws = [rng.standard_normal((F, F)) * 0.2 for _ in range(L - 1)] + [rng.standard_normal((F, Fh)) * 0.2]
    bs = [rng.standard_normal(F) * 0.1 for _ in range(L - 1)] + [rng.standard_normal(Fh) * 0.1]
    dg = rng.standard_normal((N, Fh))
    xs, g = ref_fc(x, Ws, bs, act)
    dx, dWs, dbs = ref_fc_bwd(xs, g, Ws, dg, act)

    dev = gpu_device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    tx, tW, tb, tdg = t(x), [t(w) for w in Ws], [t(b) for b in bs], t(dg)
    ty = [torch.empty(N, F, device=dev) for _ in range(L - 1)]
    tg = torch.empty(N, Fh, device=dev)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.ng_fc_block_fwd(ctx.handle, st, N, F, L, act, ptr(tx), ptr_array(tW), ptr_array(tb),
                                      ptr_array(ty), ptr(tg)), "fwd")
    for l in range(L - 1):
        np.testing.assert_allclose(ty[l].cpu().numpy(), xs[l + 1], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(tg.cpu().numpy(), g, rtol=2e-5, atol=2e-5)

    tdx = torch.empty(N, F, device=dev)
    tdW = [torch.empty_like(w) for w in tW]
    tdb = [torch.empty_like(b) for b in tb]
    scratch = torch.empty(3, N, F, device=dev)
    ctx.check(ctx.lib.ng_fc_block_bwd(ctx.handle, st, N, F, L, act, ptr_array([tx] + ty), ptr(tg), ptr_array(tW),
                                      ptr(tdg), ptr(tdx), ptr_array(tdW), ptr_array(tdb), ptr(scratch)), "bwd")
    scale = lambda a: max(1.0, np.abs(a).max())
    assert np.abs(tdx.cpu().numpy() - dx).max() < 2e-4 * scale(dx)
    for l in range(L):
        assert np.abs(tdW[l].cpu().numpy() - dWs[l]).max() < 2e-4 * scale(dWs[l]), l
        assert np.abs(tdb[l].cpu().numpy() - dbs[l]).max() < 2e-4 * scale(dbs[l]), l


def _run_block(dev, x, Ws, bs, dg, act, xs_tape=None, g_tape=None):
    """forward + backward through the C ABI; xs_tape / g_tape: feed the backward these tapes instead of the forward's"""
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr, ptr_array
    N, F = x.shape
    L, Fh = len(Ws), F // 2
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    tx, tW, tb, tdg = t(x), [t(w) for w in Ws], [t(b) for b in bs], t(dg)
    ty = [torch.empty(N, F, device=dev) for _ in range(L - 1)]
    tg = torch.empty(N, Fh, device=dev)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.ng_fc_block_fwd(ctx.handle, st, N, F, L, act, ptr(tx), ptr_array(tW), ptr_array(tb),
                                      ptr_array(ty), ptr(tg)), "fwd")
    ys = [v.cpu().numpy() for v in ty]
    g = tg.cpu().numpy()
    tape = [tx] + ty if xs_tape is None else [t(v) for v in xs_tape]
    tgt = tg if g_tape is None else t(g_tape)
    tdx = torch.empty(N, F, device=dev)
    tdW = [torch.empty_like(w) for w in tW]
    tdb = [torch.empty_like(b) for b in tb]
    scratch = torch.empty(3, N, F, device=dev)
    ctx.check(ctx.lib.ng_fc_block_bwd(ctx.handle, st, N, F, L, act, ptr_array(tape), ptr(tgt), ptr_array(tW),
                                      ptr(tdg), ptr(tdx), ptr_array(tdW), ptr_array(tdb), ptr(scratch)), "bwd")
    return ys, g, tdx.cpu().numpy(), [w.cpu().numpy() for w in tdW], [b.cpu().numpy() for b in tdb]


def _weights(rng, F, L, s=0.2):
    Fh = F // 2
    return ([rng.standard_normal((F, F)) * s for _ in range(L - 1)] + [rng.standard_normal((F, Fh)) * s],
            [rng.standard_normal(F) * 0.1 for _ in range(L - 1)] + [rng.standard_normal(Fh) * 0.1])


def test_piece_forward_redoes_a_tile_whose_activations_leave_the_fp16_range(gpu_device):
    """fc_fused.hip, piece body: activations are taken as two fp16 pieces UNSCALED; a tile with a value at or beyond 65504
    (input row, or a hidden layer's output) is redone by the same workgroup with the fp32 layers.  Rows of 3e5, a row that
    only overflows in layer 2 (6e4 growing past 65504 through the residual), inf and nan rows, next to ordinary tiles."""
    rng = np.random.default_rng(7)
    N, F, L = 1000, 64, 4
    x = rng.standard_normal((N, F))
    x[70] *= 3e5                      # tile 1: beyond the range at the input
    x[200] = 6.0e4 + 100.0 * rng.random(F)   # tile 3: inside at the input, outside after a residual layer or two
    x[900, 5] = 2.0e5                 # tile 14: one element
    Ws, bs = _weights(rng, F, L)
    Ws[0][:, :] = np.abs(Ws[0])       # positive weights: row 200 grows
    dg = rng.standard_normal((N, F // 2))
    xs, g = ref_fc(x, Ws, bs, 1)
    assert np.abs(xs[2][200]).max() > 65504 > np.abs(xs[0][200]).max()
    ys, gg, *_ = _run_block(gpu_device, x, Ws, bs, dg, 1)
    for l in range(L - 1):
        np.testing.assert_allclose(ys[l], xs[l + 1], rtol=3e-5, atol=3e-5)
    np.testing.assert_allclose(gg, g, rtol=3e-5, atol=3e-5)
    # non-finite inputs come out non-finite in their own rows and nowhere else
    x2 = x.copy()
    x2[300, 3] = np.inf
    x2[301, 4] = np.nan
    xs2, g2 = ref_fc(np.where(np.isfinite(x2), x2, 0.0), Ws, bs, 1)
    ys2, gg2, *_ = _run_block(gpu_device, x2, Ws, bs, dg, 1)
    assert not np.all(np.isfinite(ys2[0][300])) and not np.all(np.isfinite(ys2[0][301]))
    keep = np.ones(N, bool)
    keep[[300, 301]] = False
    np.testing.assert_allclose(ys2[-1][keep], xs2[-1][keep], rtol=3e-5, atol=3e-5)
    np.testing.assert_allclose(gg2[keep], g2[keep], rtol=3e-5, atol=3e-5)


@pytest.mark.parametrize("cache", [False, True])
def test_piece_kernels_take_the_fp32_body_when_a_weight_leaves_the_range(gpu_device, cache):
    """2^8 * 400 > 65504: no fp16 pieces of that weight; the pack launch raises the guard (or sets the flag word of a cached
    image) and both kernels run their fp32 bodies for the whole launch"""
    from nmrgnn_amd import _lib
    rng = np.random.default_rng(8)
    N, F, L = 777, 64, 3
    x = rng.standard_normal((N, F)) * 0.1
    Ws, bs = _weights(rng, F, L, s=0.05)
    Ws[1][3, 5] = 400.0
    dg = rng.standard_normal((N, F // 2))
    xs, g = ref_fc(x, Ws, bs, 1)
    dx, dWs, dbs = ref_fc_bwd(xs, g, Ws, dg, 1)
    ctx = _lib.get_context(0)
    if cache:
        ctx.lib.ng_weights_frozen(ctx.handle, 12345)
    try:
        for _ in range(2 if cache else 1):      # second round: the cached image and its flag word
            ys, gg, tdx, tdW, tdb = _run_block(gpu_device, x, Ws, bs, dg, 1)
            np.testing.assert_allclose(ys[-1], xs[-1], rtol=3e-5, atol=3e-5)
            np.testing.assert_allclose(gg, g, rtol=3e-5, atol=3e-5)
            sc = lambda a: max(1.0, np.abs(a).max())
            assert np.abs(tdx - dx).max() < 1e-4 * sc(dx)
            for l in range(L):
                assert np.abs(tdW[l] - dWs[l]).max() < 1e-4 * sc(dWs[l]), l
                assert np.abs(tdb[l] - dbs[l]).max() < 1e-4 * sc(dbs[l]), l
    finally:
        if cache:
            ctx.lib.ng_weights_frozen(ctx.handle, 0)


def test_piece_backward_with_gradient_rows_spanning_decades_and_large_inputs(gpu_device, monkeypatch):
    """The dP rows go into the fp16 planes with a power-of-two scale of their own (a labelled atom's gradient next to rows 1e-6
    of it, and all-zero rows), the x operand of the dW product takes the inverse and — for a feature column with entries
    beyond 2^15 — a column scale.  Every gradient against float64, to 2e-5 of the tensor's largest entry and no worse than
    8 x the f32-input kernels on the same inputs (2^-21 against 2^-24; + rounding floor)."""
    rng = np.random.default_rng(9)
    N, F, L = 1500, 64, 4
    x = rng.standard_normal((N, F))
    x[:, 7] *= 4.0e4                          # a feature column beyond 2^15 in the tape
    x[40:50] *= 300.0
    Ws, bs = _weights(rng, F, L, s=0.1)
    dg = rng.standard_normal((N, F // 2)) * 10.0 ** rng.uniform(-6, 0, (N, 1))
    dg[rng.random(N) < 0.3] = 0.0
    dg[5] *= 1e4
    xs, g = ref_fc(x, Ws, bs, 1)
    dx, dWs, dbs = ref_fc_bwd(xs, g, Ws, dg, 1)
    res = {}
    for mode in ("f16x2", "fp32"):
        monkeypatch.setenv("NG_GEMM_MATH", mode)
        # the float64 tapes (rounded to fp32) for both modes: the backward is compared on identical inputs
        res[mode] = _run_block(gpu_device, x, Ws, bs, dg, 1, xs_tape=xs, g_tape=g)[2:]
    ref = [dx] + dWs + dbs
    names = ["dx"] + ["dW%d" % l for l in range(L)] + ["db%d" % l for l in range(L)]
    flat = lambda r: [r[0]] + list(r[1]) + list(r[2])
    for name, want, got2, got1 in zip(names, ref, flat(res["f16x2"]), flat(res["fp32"])):
        mx = np.abs(want).max()
        e2, e1 = np.abs(got2 - want).max() / mx, np.abs(got1 - want).max() / mx
        assert e2 < 2e-5 and e2 <= 8.0 * e1 + 2e-6, (name, e2, e1)
    # rows with tiny upstream gradients keep their relative accuracy in dx (own scale per row)
    small = np.nonzero((np.abs(dg).max(1) > 0) & (np.abs(dg).max(1) < 1e-4))[0]
    got = flat(res["f16x2"])[0]
    for i in small[:50]:
        assert np.abs(got[i] - dx[i]).max() <= 2e-5 * np.abs(dx[i]).max() + 1e-30, i
