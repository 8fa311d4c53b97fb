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
