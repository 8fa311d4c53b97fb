"""ng_dense_fwd / ng_dense_bwd on the split-operand tile GEMMs (gemm_h2.hip: two fp16 pieces per fp32 operand, three
piece products; shapes K % 32 == 0, N % 128 == 0, M >= 4096) against float64
and against the f32-input MFMA GEMM (NG_GEMM_MATH=fp32): bias, activation, residual, saved activation, ragged M."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


@pytest.mark.parametrize("M,K,N,act,residual", [(4096, 128, 128, 1, 0), (5000, 256, 256, 1, 1), (138500, 768, 256, 0, 0),
                                                (4097, 64, 128, 0, 0), (300000, 128, 128, 1, 0),
                                                # one molecule per call: the 32-row short-operand kernel
                                                (2770, 768, 256, 1, 0), (300, 256, 256, 1, 1), (2771, 128, 256, 0, 0),
                                                (2770, 384, 512, 1, 0)])
def test_dense_fwd_split_vs_float64(gpu_device, monkeypatch, M, K, N, act, residual):
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    rng = np.random.default_rng(M + K)
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((K, N)) * 0.1).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    pre = X.astype(np.float64) @ W.astype(np.float64) + b
    s_ref = softplus(pre) if act else pre
    y_ref = s_ref + (X if residual else 0)
    mag = (np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64)).max()
    out = {}
    for math in ("f16x2", "fp32"):
        monkeypatch.setenv("NG_GEMM_MATH", math)
        tX, tW, tb = (torch.from_numpy(a).to(gpu_device) for a in (X, W, b))
        Y = torch.full((M, N), 7.0, device=gpu_device)
        S = torch.full((M, N), 7.0, device=gpu_device)
        ctx = _lib.get_context(0)
        st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
        ctx.check(ctx.lib.ng_dense_fwd(ctx.handle, st, M, K, N, act, residual, ptr(tX), ptr(tW), ptr(tb), ptr(Y), ptr(S)),
                  "ng_dense_fwd")
        torch.cuda.synchronize()
        out[math] = (Y.cpu().numpy().astype(np.float64), S.cpu().numpy().astype(np.float64))
    for k, (y, s) in out.items():
        assert np.isfinite(y).all(), k
        assert np.abs(y - y_ref).max() < 2e-6 * mag, k
        assert np.abs(s - s_ref).max() < 2e-6 * mag, k
    e3 = np.sqrt(((out["f16x2"][0] - y_ref) ** 2).mean())
    e1 = np.sqrt(((out["fp32"][0] - y_ref) ** 2).mean())
    assert e3 < 1.5 * e1 + 1e-8, (e3, e1)


@pytest.mark.parametrize("M,K,N,act,residual", [(4096, 128, 128, 1, 0), (5000, 256, 256, 1, 1), (70000, 256, 768, 0, 0)])
def test_dense_bwd_dx_split_vs_float64(gpu_device, monkeypatch, M, K, N, act, residual):
    """dX of ng_dense_bwd (dP = dY * act'(s); dX = (residual ? dY : 0) + dP W^T) on the split-operand GEMMs (dW: the row-contracting variant with
    transposing LDS reads); db stays a column-sum kernel"""
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    if residual:
        N = K
    rng = np.random.default_rng(M + N)
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((K, N)) * 0.1).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    dY = rng.standard_normal((M, N)).astype(np.float32)
    pre = X.astype(np.float64) @ W.astype(np.float64) + b
    s = (softplus(pre) if act else pre).astype(np.float32)
    dP = dY.astype(np.float64) * ((1.0 - np.exp(-s.astype(np.float64))) if act else 1.0)
    dX_ref = dP @ W.astype(np.float64).T + (dY if residual else 0)
    dW_ref = X.astype(np.float64).T @ dP
    mag = (np.abs(dP) @ np.abs(W.astype(np.float64)).T).max()
    res = {}
    for math in ("f16x2", "fp32"):
        monkeypatch.setenv("NG_GEMM_MATH", math)
        tX, tW, tS, tdY = (torch.from_numpy(a).to(gpu_device) for a in (X, W, s, dY))
        dX = torch.full((M, K), 7.0, device=gpu_device)
        dW = torch.full((K, N), 7.0, device=gpu_device)
        db = torch.full((N,), 7.0, device=gpu_device)
        ctx = _lib.get_context(0)
        st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
        ctx.check(ctx.lib.ng_dense_bwd(ctx.handle, st, M, K, N, act, residual, ptr(tX), ptr(tW), ptr(tS), ptr(tdY), ptr(dX),
                                       ptr(dW), ptr(db)), "ng_dense_bwd")
        torch.cuda.synchronize()
        res[math] = dX.cpu().numpy().astype(np.float64)
        assert np.isfinite(res[math]).all()
        assert np.abs(res[math] - dX_ref).max() < 2e-6 * mag, math
        magw = (np.abs(X).astype(np.float64).T @ np.abs(dP)).max()
        assert np.abs(dW.cpu().numpy() - dW_ref).max() < 2e-6 * magw, math
        assert np.abs(db.cpu().numpy() - dP.sum(0)).max() < 2e-6 * np.abs(dP).sum(0).max(), math
    e3 = np.sqrt(((res["f16x2"] - dX_ref) ** 2).mean())
    e1 = np.sqrt(((res["fp32"] - dX_ref) ** 2).mean())
    assert e3 < 1.5 * e1 + 1e-8, (e3, e1)


@pytest.mark.parametrize("shift", [-60, -24, -7, 11, 40])
def test_dense_bwd_gradient_scale_is_exact(gpu_device, monkeypatch, shift):
    """The fp16-piece GEMMs split a gradient operand as S * dP with a power of two S taken from max|dY| (gemm_grad_scale),
    so that tiny gradients (loss scaling, 1/N) neither underflow the pieces nor large ones overflow them.  Property:
    dY * 2^shift gives dX * 2^shift and dW * 2^shift bit for bit."""
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    M, K, N, act = 5000, 256, 256, 1
    rng = np.random.default_rng(5)
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((K, N)) * 0.1).astype(np.float32)
    s = softplus(X.astype(np.float64) @ W.astype(np.float64)).astype(np.float32)
    dY = rng.standard_normal((M, N)).astype(np.float32)
    monkeypatch.setenv("NG_GEMM_MATH", "f16x2")
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)

    def run(dy):
        tX, tW, tS, tdY = (torch.from_numpy(a).to(gpu_device) for a in (X, W, s, dy))
        dX = torch.empty((M, K), device=gpu_device)
        dW = torch.empty((K, N), device=gpu_device)
        db = torch.empty((N,), device=gpu_device)
        ctx.check(ctx.lib.ng_dense_bwd(ctx.handle, st, M, K, N, act, 0, ptr(tX), ptr(tW), ptr(tS), ptr(tdY), ptr(dX),
                                       ptr(dW), ptr(db)), "ng_dense_bwd")
        torch.cuda.synchronize()
        return dX.cpu().numpy().astype(np.float64), dW.cpu().numpy().astype(np.float64)

    base = run(dY)
    moved = run((dY.astype(np.float64) * 2.0 ** shift).astype(np.float32))
    for a, b in zip(moved, base):
        assert np.isfinite(a).all()
        assert np.array_equal(a, b * 2.0 ** shift)


@pytest.mark.parametrize("M,K,N", [(5000, 256, 256), (2770, 768, 256), (70000, 256, 768)])
def test_dense_operands_beyond_the_fp16_range_equal_float64(gpu_device, monkeypatch, M, K, N):
    """The reference's Dense is plain fp32 (nmrgnn/model.py:191-196): a feature of 3e5 is an ordinary number.  The
    split-operand GEMMs hold |x| < 65504 per piece; an operand beyond that raises the range guard and the entry point
    re-runs the product on f32-input MFMA by itself (ng_internal.h: RangeGuard) — forward, dX and dW equal float64 and
    equal the NG_GEMM_MATH=fp32 bits."""
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    rng = np.random.default_rng(M + 3)
    X = rng.standard_normal((M, K)).astype(np.float32)
    X[rng.integers(0, M, 50), rng.integers(0, K, 50)] = 3.0e5          # a few features far outside the fp16 range
    W = (rng.standard_normal((K, N)) * 0.1).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    dY = rng.standard_normal((M, N)).astype(np.float32)
    X64, W64 = X.astype(np.float64), W.astype(np.float64)
    y_ref = X64 @ W64 + b
    dX_ref = dY.astype(np.float64) @ W64.T
    dW_ref = X64.T @ dY.astype(np.float64)
    ctx = _lib.get_context(0)
    res = {}
    for math in ("f16x2", "fp32"):
        monkeypatch.setenv("NG_GEMM_MATH", math)
        tX, tW, tb, tdY = (torch.from_numpy(a).to(gpu_device) for a in (X, W, b, dY))
        Y = torch.full((M, N), 7.0, device=gpu_device)
        dX = torch.full((M, K), 7.0, device=gpu_device)
        dW = torch.full((K, N), 7.0, device=gpu_device)
        db = torch.full((N,), 7.0, device=gpu_device)
        st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
        ctx.check(ctx.lib.ng_dense_fwd(ctx.handle, st, M, K, N, 0, 0, ptr(tX), ptr(tW), ptr(tb), ptr(Y), None), "fwd")
        ctx.check(ctx.lib.ng_dense_bwd(ctx.handle, st, M, K, N, 0, 0, ptr(tX), ptr(tW), None, ptr(tdY), ptr(dX), ptr(dW),
                                       ptr(db)), "bwd")
        torch.cuda.synchronize()
        res[math] = [t.cpu().numpy().astype(np.float64) for t in (Y, dX, dW)]
    for math, (y, dx, dw) in res.items():
        for got, ref, mag in ((y, y_ref, np.abs(X64) @ np.abs(W64)), (dx, dX_ref, np.abs(dY.astype(np.float64)) @ np.abs(W64).T),
                              (dw, dW_ref, np.abs(X64).T @ np.abs(dY.astype(np.float64)))):
            assert np.isfinite(got).all(), math
            assert np.abs(got - ref).max() < 3e-6 * mag.max(), math
    # Y sees the out-of-range operand X: its fallback is the f32-input kernel, bit for bit (dW too, but its row split
    # follows the split-operand tiling, so the partial sums differ in the last bits; dX = dY W^T has no such operand)
    np.testing.assert_array_equal(res["f16x2"][0], res["fp32"][0])
