"""Edge forward / backward on the fp16 matrix pipe with two-piece split fp32 operands (edge_fwd_h2.hip, edge_bwd_h2.hip:
three piece products per multiply, fp32 accumulate) against a float64 numpy statement of nmrgnn/model.py:251-261 +
layers.py:137-140 + model.py:132-138 — and against the f32-input MFMA kernels (NG_EDGE_MATH=fp32): the split path must be
as close to float64 as the fp32 path is."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
H = 128


def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def ref_edge(d_src, d_eff, centers, gap, Ws, bs):
    m = (d_src > 0).astype(np.float64)
    x = np.exp(-(d_eff[:, None] - centers[None, :]) ** 2 / gap) * m[:, None]
    zs = []
    for W, b in zip(Ws[:-1], bs[:-1]):
        x = softplus(x @ W + b)
        zs.append(x)
    return m[:, None] * (x @ Ws[-1] + bs[-1]), zs


def tape_perm(n):
    """index map of the blocked tape: blocked_flat[perm] == row-major flat, for one layer [n][128]"""
    idx = np.arange(n * H, dtype=np.int64).reshape(n, H)            # row-major positions
    out = idx.copy().reshape(-1)
    G = n // 32
    if G:
        r = np.arange(32)[:, None]
        f = np.arange(H)[None, :]
        bo, q, hf, j = f // 32, (f % 32) // 8, (f % 8) // 4, f % 4
        within = ((bo * 4 + q) * 64 + hf * 32 + r) * 4 + j          # [32][128] position inside a group
        pos = (np.arange(G)[:, None, None] * 4096 + within[None]).reshape(G * 32, H)
        out = np.concatenate([pos.reshape(-1), idx[G * 32:].reshape(-1)])
    return out                                                      # out[row*H + col] = position in the tape


def tape_layout(E, n):
    from nmrgnn_amd import _lib
    return int(_lib.get_context(0).lib.ng_edge_tape_layout(H, E, 4, 1, n))


def run_gpu(dev, d_src, d_eff, centers, gap, Ws, bs, E, save):
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr, ptr_array
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    n = len(d_src)
    td, te, tc = t(d_src), t(d_eff), t(centers)
    tW, tb = [t(w) for w in Ws], [t(b) for b in bs]
    e = torch.full((n, E), 7.0, device=dev)
    z = torch.full((3, n, H), 7.0, device=dev) if save else None
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.ng_edge_mlp_fwd(ctx.handle, st, n, H, E, 4, 1, ptr(td), ptr(te), ptr(tc), float(gap), ptr_array(tW),
                                      ptr_array(tb), ptr(e), ptr(z)), "ng_edge_mlp_fwd")
    torch.cuda.synchronize()
    zz = None
    if save:
        zz = z.cpu().numpy().astype(np.float64)
        if tape_layout(E, n):                        # blocked tape -> row-major for the comparison
            perm = tape_perm(n)
            zz = np.stack([zz[l].reshape(-1)[perm].reshape(n, H) for l in range(3)])
    return e.cpu().numpy().astype(np.float64), zz


@pytest.mark.parametrize("n,E,save", [(1, 3, True), (255, 3, True), (257, 1, True), (70001, 3, True), (5000, 8, False),
                                      (4096, 3, False), (33, 2, True)])
def test_edge_forward_split_vs_float64(gpu_device, monkeypatch, n, E, save):
    rng = np.random.default_rng(n + E)
    d_src = rng.uniform(0.05, 1.2, n)
    d_src[rng.random(n) < 0.15] = 0.0                       # padded slots
    d_eff = np.where(d_src > 0, d_src + 0.025 * rng.standard_normal(n), d_src)
    centers = np.linspace(0.0, 1.2, H)
    gap = centers[1] - centers[0]
    Ws = [rng.standard_normal((H, H)) * 0.15 for _ in range(3)] + [rng.standard_normal((H, E)) * 0.2]
    bs = [rng.standard_normal(H) * 0.1 for _ in range(3)] + [rng.standard_normal(E) * 0.1]
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)     # what the GPU is given
    e_ref, z_ref = ref_edge(f32(d_src), f32(d_eff), f32(centers), float(np.float32(gap)), [f32(w) for w in Ws],
                            [f32(b) for b in bs])
    out = {}
    for math in ("f16x2", "fp32"):
        monkeypatch.setenv("NG_EDGE_MATH", math)
        out[math] = run_gpu(gpu_device, d_src, d_eff, centers, gap, Ws, bs, E, save)
    # the output layer is a 128-term dot product: its rounding error scales with sum |z||W|, not with the result
    mag = (np.abs(z_ref[2]) @ np.abs(f32(Ws[3])) + np.abs(f32(bs[3]))).max()
    err = {k: np.abs(v[0] - e_ref) for k, v in out.items()}
    for split in ("f16x2",):
        assert err[split].max() < 1e-6 * mag, (split, err[split].max(), err["fp32"].max(), mag)
        assert np.sqrt((err[split] ** 2).mean()) < 2e-7 * mag, split
        if save:
            # hidden activations: the split products carry no more error than the f32-input MFMA chain
            for l in range(3):
                dz = {k: v[1][l] - z_ref[l] for k, v in out.items()}
                assert np.abs(dz[split]).max() < 1e-5, (split, l)
                if n >= 255:
                    rms = {k: np.sqrt((v ** 2).mean()) for k, v in dz.items()}
                    assert rms[split] < 1.2 * rms["fp32"] + 1e-8, (split, l, rms)
        # masked edges give exact zeros, rows past the end are never written
        assert np.all(out[split][0][d_src == 0] == 0.0)


def ref_edge_bwd(d_src, d_eff, centers, gap, Ws, bs, de):
    """float64 backward of ref_edge w.r.t. the weights and biases (distances are not trainable)"""
    m = (d_src > 0).astype(np.float64)
    R = np.exp(-(d_eff[:, None] - centers[None, :]) ** 2 / gap) * m[:, None]
    xs = [R]
    for W, b in zip(Ws[:-1], bs[:-1]):
        xs.append(softplus(xs[-1] @ W + b))
    dE = de * m[:, None]
    dWs, dbs = [None] * 4, [None] * 4
    dWs[3], dbs[3] = xs[3].T @ dE, dE.sum(0)
    g = dE @ Ws[3].T
    for l in (2, 1, 0):
        G = g * (1.0 - np.exp(-xs[l + 1]))
        dWs[l], dbs[l] = xs[l].T @ G, G.sum(0)
        g = G @ Ws[l].T
    return dWs, dbs, xs[1:]


def run_gpu_bwd(dev, d_src, d_eff, centers, gap, Ws, zs, de, E):
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr, ptr_array
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    n = len(d_src)
    td, te, tc, tde = t(d_src), t(d_eff), t(centers), t(de)
    tW = [t(w) for w in Ws]
    zs = np.stack(zs)
    if tape_layout(E, n):                            # hand the kernel the tape in the layout it expects
        perm = tape_perm(n)
        blk = np.empty_like(zs).reshape(3, -1)
        for l in range(3):
            blk[l][perm] = zs[l].reshape(-1)
        zs = blk.reshape(3, n, H)
    tz = t(zs)
    dW = [torch.full((H, H), 7.0, device=dev) for _ in range(3)] + [torch.full((H, E), 7.0, device=dev)]
    db = [torch.full((H,), 7.0, device=dev) for _ in range(3)] + [torch.full((E,), 7.0, device=dev)]
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.ng_edge_mlp_bwd(ctx.handle, st, n, H, E, 4, 1, ptr(td), ptr(te), ptr(tc), float(gap), ptr_array(tW),
                                      ptr(tz), ptr(tde), ptr_array(dW), ptr_array(db)), "ng_edge_mlp_bwd")
    torch.cuda.synchronize()
    return [w.cpu().numpy().astype(np.float64) for w in dW], [b.cpu().numpy().astype(np.float64) for b in db]


@pytest.mark.parametrize("n,E", [(1, 3), (63, 3), (65, 1), (5000, 4), (70001, 3), (4096, 2), (3000, 8), (1048576, 3)])
def test_edge_backward_split_vs_float64(gpu_device, monkeypatch, n, E):
    """weight / bias gradients of the edge path: split-operand kernel (default, E <= 4; E = 8 exercises the fall-back)
    and the f32-input MFMA kernel against float64, on the SAME saved activations (float32-rounded float64 ones)"""
    rng = np.random.default_rng(7 * n + E)
    d_src = rng.uniform(0.05, 1.2, n)
    d_src[rng.random(n) < 0.15] = 0.0
    d_eff = np.where(d_src > 0, d_src + 0.025 * rng.standard_normal(n), d_src)
    centers = np.linspace(0.0, 1.2, H)
    gap = centers[1] - centers[0]
    Ws = [rng.standard_normal((H, H)) * 0.15 for _ in range(3)] + [rng.standard_normal((H, E)) * 0.2]
    bs = [rng.standard_normal(H) * 0.1 for _ in range(3)] + [rng.standard_normal(E) * 0.1]
    de = rng.standard_normal((n, E))
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    Wf, bf = [f32(w) for w in Ws], [f32(b) for b in bs]
    _, _, zs = ref_edge_bwd(f32(d_src), f32(d_eff), f32(centers), float(np.float32(gap)), Wf, bf, f32(de))
    zs32 = [f32(z) for z in zs]            # what the forward would have saved
    # reference gradients computed FROM the float32-rounded activations, like the kernels do
    m = (f32(d_src) > 0).astype(np.float64)
    R = np.exp(-(f32(d_eff)[:, None] - f32(centers)[None, :]) ** 2 / float(np.float32(gap))) * m[:, None]
    xs = [R] + zs32
    dE = f32(de) * m[:, None]
    # a gradient entry is a sum over the edges: its rounding error scales with the sum of |terms| (mag), not with the
    # (often cancelling) result
    ref_dW, ref_db, mag_dW, mag_db = [None] * 4, [None] * 4, [None] * 4, [None] * 4
    ref_dW[3], ref_db[3] = xs[3].T @ dE, dE.sum(0)
    mag_dW[3], mag_db[3] = np.abs(xs[3]).T @ np.abs(dE), np.abs(dE).sum(0)
    g = dE @ Wf[3].T
    for l in (2, 1, 0):
        G = g * (1.0 - np.exp(-xs[l + 1]))
        ref_dW[l], ref_db[l] = xs[l].T @ G, G.sum(0)
        mag_dW[l], mag_db[l] = np.abs(xs[l]).T @ np.abs(G), np.abs(G).sum(0)
        g = G @ Wf[l].T
    out = {}
    for math in ("f16x2", "fp32"):
        monkeypatch.setenv("NG_EDGE_MATH", math)
        out[math] = run_gpu_bwd(gpu_device, d_src, d_eff, centers, gap, Ws, zs32, de, E)
    for l in range(4):
        for name, ref, mag, k in (("dW", ref_dW[l], mag_dW[l], 0), ("db", ref_db[l], mag_db[l], 1)):
            scale = max(mag.max(), 1e-6)
            err = {mth: np.abs(o[k][l] - ref).max() / scale for mth, o in out.items()}
            for split in ("f16x2",):
                assert err[split] < 1e-6, (split, name, l, err)                 # ~16 fp32 roundings of the term scale
                assert err[split] < 4.0 * err["fp32"] + 1e-7, (split, name, l, err)


@pytest.mark.parametrize("shift", [-40, -13, 9, 30])
def test_edge_backward_f16x2_gradient_scale_is_exact(gpu_device, monkeypatch, shift):
    """The two-piece fp16 backward runs on S * dE with S a power of two picked per call from max|dE| and the weight
    norms (edge_bwd_h2.hip), so that fp16 pieces neither overflow nor lose bits whatever the loss scaling is.
    Size-independent property: multiplying dE by 2^shift must multiply every gradient by exactly 2^shift — bit for bit
    (S moves by the same power of two; nothing else sees the change)."""
    n, E = 5000, 3
    rng = np.random.default_rng(11)
    d_src = rng.uniform(0.05, 1.2, n)
    d_src[rng.random(n) < 0.15] = 0.0
    centers = np.linspace(0.0, 1.2, H)
    gap = centers[1] - centers[0]
    Ws = [rng.standard_normal((H, H)) * 0.15 for _ in range(3)] + [rng.standard_normal((H, E)) * 0.2]
    bs = [rng.standard_normal(H) * 0.1 for _ in range(3)] + [rng.standard_normal(E) * 0.1]
    de = rng.standard_normal((n, E)).astype(np.float32).astype(np.float64)
    _, _, zs = ref_edge_bwd(d_src, d_src, centers, gap, Ws, bs, de)
    zs32 = [np.asarray(z, dtype=np.float32).astype(np.float64) for z in zs]
    monkeypatch.setenv("NG_EDGE_MATH", "f16x2")
    base = run_gpu_bwd(gpu_device, d_src, d_src, centers, gap, Ws, zs32, de, E)
    moved = run_gpu_bwd(gpu_device, d_src, d_src, centers, gap, Ws, zs32, de * 2.0 ** shift, E)
    for k in range(2):
        for l in range(4):
            assert np.array_equal(moved[k][l], base[k][l] * 2.0 ** shift), (k, l)
    assert all(np.isfinite(g).all() for k in range(2) for g in moved[k])


def test_edge_backward_beyond_one_launch_segment(gpu_device):
    """Edge lists longer than 2^23 - 256 edges (the 32-bit buffer offsets of one split-operand launch: the 1-GPU point
    of a 4096-graph strong-scaling run has 16.8 M edges) run as several launches of the SAME kernel.  Size-independent
    property: the weight gradients are sums over edges, so the gradients of the whole list equal the sum of the
    gradients of its two halves (each below the segment size), and the tape layout stays blocked."""
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr, ptr_array
    dev = gpu_device
    n = (1 << 23) + 70001                       # 2 segments, ragged last tile and last 32-group
    E = 3
    assert tape_layout(E, n) == 1
    g = torch.Generator(device="cpu").manual_seed(1)
    d = (torch.rand(n, generator=g) * 0.45 + 0.05)
    d[torch.rand(n, generator=g) < 0.1] = 0.0
    rng = np.random.default_rng(3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    centers = np.linspace(0.005, 0.2, H).astype(np.float32)
    gap = float(centers[1] - centers[0])
    Ws = [t(rng.standard_normal((H, H)) * 0.15) for _ in range(3)] + [t(rng.standard_normal((H, E)) * 0.2)]
    bs = [t(rng.standard_normal(H) * 0.1) for _ in range(3)] + [t(rng.standard_normal(E) * 0.1)]
    td, tc = d.to(dev), t(centers)
    tde = torch.randn(n, E, generator=g).to(dev)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def fwd_bwd(lo, hi):
        m = hi - lo
        e = torch.empty(m, E, device=dev)
        z = torch.empty(3, m, H, device=dev)
        dd, de = td[lo:hi].contiguous(), tde[lo:hi].contiguous()
        ctx.check(ctx.lib.ng_edge_mlp_fwd(ctx.handle, st, m, H, E, 4, 1, ptr(dd), ptr(dd), ptr(tc), gap, ptr_array(Ws),
                                          ptr_array(bs), ptr(e), ptr(z)), "fwd")
        dW = [torch.empty(H, H, device=dev) for _ in range(3)] + [torch.empty(H, E, device=dev)]
        db = [torch.empty(H, device=dev) for _ in range(3)] + [torch.empty(E, device=dev)]
        ctx.check(ctx.lib.ng_edge_mlp_bwd(ctx.handle, st, m, H, E, 4, 1, ptr(dd), ptr(dd), ptr(tc), gap, ptr_array(Ws),
                                          ptr(z), ptr(de), ptr_array(dW), ptr_array(db)), "bwd")
        torch.cuda.synchronize()
        return e, [w.double() for w in dW], [b.double() for b in db]

    cut = 4 * 1024 * 1024 + 96                    # a multiple of 32: both halves keep whole tape groups
    e_all, dW_all, db_all = fwd_bwd(0, n)
    e_a, dW_a, db_a = fwd_bwd(0, cut)
    e_b, dW_b, db_b = fwd_bwd(cut, n)
    assert torch.equal(e_all[:cut], e_a) and torch.equal(e_all[cut:], e_b)          # per-edge results do not move
    for l in range(4):
        for whole, a, b in ((dW_all[l], dW_a[l], dW_b[l]), (db_all[l], db_a[l], db_b[l])):
            ref = a + b
            scale = max(float(a.abs().max() + b.abs().max()), 1e-6)
            # fp32 accumulation over 8.5 M edges in a different partition (512 instead of 2 x 256 partial sums)
            assert float((whole - ref).abs().max()) / scale < 2e-5, l


@pytest.mark.parametrize("wscale", [0.003, 1.0, 8.0])
def test_edge_forward_split_across_weight_magnitudes(gpu_device, monkeypatch, wscale):
    """The fp16 pieces are taken from 2^8 W and from unscaled activations: weights 300 times smaller than Glorot's (their
    l pieces become fp16 subnormals, which the matrix pipe honours) and 8 times larger (a gain of ~14 per layer, activations
    in the thousands, still below 65504) must keep the float64 agreement of the default case, relative to the size of
    the summed terms."""
    n, E = 4096, 3
    rng = np.random.default_rng(17)
    d_src = rng.uniform(0.05, 1.2, n)
    d_src[rng.random(n) < 0.1] = 0.0
    centers = np.linspace(0.0, 1.2, H)
    gap = centers[1] - centers[0]
    Ws = [rng.standard_normal((H, H)) * 0.15 * wscale for _ in range(3)] + [rng.standard_normal((H, E)) * 0.2 * wscale]
    bs = [rng.standard_normal(H) * 0.1 * wscale for _ in range(3)] + [rng.standard_normal(E) * 0.1]
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    e_ref, z_ref = ref_edge(f32(d_src), f32(d_src), f32(centers), float(np.float32(gap)), [f32(w) for w in Ws],
                            [f32(b) for b in bs])
    out = {}
    for math in ("f16x2", "fp32"):
        monkeypatch.setenv("NG_EDGE_MATH", math)
        out[math] = run_gpu(gpu_device, d_src, d_src, centers, gap, Ws, bs, E, True)
    assert np.isfinite(out["f16x2"][0]).all()
    mag = (np.abs(z_ref[2]) @ np.abs(f32(Ws[3])) + np.abs(f32(bs[3]))).max()
    err = {k: np.abs(v[0] - e_ref).max() for k, v in out.items()}
    # the error of a deep chain grows with the weights' gain (each layer amplifies the previous layer's rounding); the
    # split path has to stay within twice the f32-input MFMA path's error, whatever that gain is
    assert err["f16x2"] < 2.0 * err["fp32"] + 1e-6 * mag, (wscale, err, mag)
    for l in range(3):
        zmag = max(np.abs(z_ref[l]).max(), 1.0)
        dz = {k: np.abs(v[1][l] - z_ref[l]).max() / zmag for k, v in out.items()}
        assert dz["f16x2"] < 2.0 * dz["fp32"] + 5e-7, (wscale, l, dz)


def _overflow_case(n=1024, E=3, seed=18, wscale=6.0):
    rng = np.random.default_rng(seed)
    d_src = rng.uniform(0.05, 1.2, n)
    d_src[rng.random(n) < 0.1] = 0.0
    centers = np.linspace(0.0, 1.2, H)
    gap = centers[1] - centers[0]
    # weights 40 times Glorot's: a gain of ~68 per layer, third-layer activations ~1e5 — beyond the 65504 of an fp16 piece
    Ws = [rng.standard_normal((H, H)) * wscale for _ in range(3)] + [rng.standard_normal((H, E)) * 0.2]
    bs = [rng.standard_normal(H) * 0.1 for _ in range(3)] + [rng.standard_normal(E) * 0.1]
    return d_src, centers, gap, Ws, bs


@pytest.mark.parametrize("n", [1024, 1000])          # full 32-edge tape groups only / with a row-major tail group
def test_edge_forward_beyond_the_fp16_range_equals_float64(gpu_device, monkeypatch, n):
    """The reference's Dense layers are plain fp32 (nmrgnn/model.py:132-138): an activation of 1e5 is an ordinary number.
    The default kernels split operands into fp16 pieces (|x| < 65504); when an operand leaves that range the kernel
    raises the range guard and the entry point re-runs the call on f32-input MFMA by itself (ng_internal.h:
    RangeGuard) — outputs AND the saved-activation tape (in the layout the split-operand kernel would have written)
    equal the float64 statement, as they do with NG_EDGE_MATH=fp32."""
    E = 3
    d_src, centers, gap, Ws, bs = _overflow_case(n, E)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    e_ref, z_ref = ref_edge(f32(d_src), f32(d_src), f32(centers), float(np.float32(gap)), [f32(w) for w in Ws], [f32(b) for b in bs])
    assert np.abs(z_ref[2]).max() > 65504                         # the case really leaves the fp16 range
    out = {}
    for math in ("f16x2", "fp32"):
        monkeypatch.setenv("NG_EDGE_MATH", math)
        out[math] = run_gpu(gpu_device, d_src, d_src, centers, gap, Ws, bs, E, True)
    tol = 1e-4 * np.abs(e_ref).max()
    for math in ("f16x2", "fp32"):
        e, z = out[math]
        assert np.isfinite(e).all() and np.isfinite(z).all(), math
        assert np.abs(e - e_ref).max() < tol, math
        for l in range(3):
            assert np.abs(z[l] - z_ref[l]).max() < 1e-5 * np.abs(z_ref[l]).max(), (math, l)
    # the fallback IS the f32-input kernel: same bits as the explicit switch
    np.testing.assert_array_equal(out["f16x2"][0], out["fp32"][0])
    assert np.all(out["f16x2"][0][d_src == 0] == 0.0)
    # and an in-range call right after it is back on the split-operand kernels (the guard is per call)
    monkeypatch.setenv("NG_EDGE_MATH", "f16x2")
    Wn = [w * (0.15 / 6.0) for w in Ws[:3]] + [Ws[3]]
    e2, _ = run_gpu(gpu_device, d_src, d_src, centers, gap, Wn, bs, E, False)
    monkeypatch.setenv("NG_EDGE_MATH", "fp32")
    e3, _ = run_gpu(gpu_device, d_src, d_src, centers, gap, Wn, bs, E, False)
    assert np.isfinite(e2).all() and not np.array_equal(e2, e3)     # different arithmetic, both fine
    assert np.abs(e2 - e3).max() < 1e-5 * max(np.abs(e3).max(), 1.0)


def test_edge_backward_beyond_the_fp16_range_equals_float64(gpu_device, monkeypatch):
    """the same for the weight / bias gradients.  The backward splits Z1, Z2 (dW operands) and the weights into fp16 pieces
    (Z3 only enters fp32 VALU work), so the case needs SECOND-layer activations beyond 65504: weights 200 times Glorot's"""
    n, E = 3000, 3
    d_src, centers, gap, Ws, bs = _overflow_case(n, E, seed=4, wscale=30.0)
    rng = np.random.default_rng(2)
    de = rng.standard_normal((n, E))
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    Wf, bf = [f32(w) for w in Ws], [f32(b) for b in bs]
    _, _, zs = ref_edge_bwd(f32(d_src), f32(d_src), f32(centers), float(np.float32(gap)), Wf, bf, f32(de))
    zs32 = [f32(z) for z in zs]
    assert np.abs(zs32[1]).max() > 65504 and np.isfinite(zs32[2]).all()
    m = (f32(d_src) > 0).astype(np.float64)
    R = np.exp(-(f32(d_src)[:, None] - f32(centers)[None, :]) ** 2 / float(np.float32(gap))) * m[:, None]
    xs = [R] + zs32
    dE = f32(de) * m[:, None]
    ref_dW, ref_db, mag_dW, mag_db = [None] * 4, [None] * 4, [None] * 4, [None] * 4
    ref_dW[3], ref_db[3] = xs[3].T @ dE, dE.sum(0)
    mag_dW[3], mag_db[3] = np.abs(xs[3]).T @ np.abs(dE), np.abs(dE).sum(0)
    g = dE @ Wf[3].T
    for l in (2, 1, 0):
        G = g * (1.0 - np.exp(-xs[l + 1]))
        ref_dW[l], ref_db[l] = xs[l].T @ G, G.sum(0)
        mag_dW[l], mag_db[l] = np.abs(xs[l]).T @ np.abs(G), np.abs(G).sum(0)
        g = G @ Wf[l].T
    out = {}
    for math in ("f16x2", "fp32"):
        monkeypatch.setenv("NG_EDGE_MATH", math)
        out[math] = run_gpu_bwd(gpu_device, d_src, d_src, centers, gap, Ws, zs32, de, E)
    for l in range(4):
        for ref, mag, k in ((ref_dW[l], mag_dW[l], 0), (ref_db[l], mag_db[l], 1)):
            scale = max(mag.max(), 1e-6)
            for math in ("f16x2", "fp32"):
                got = out[math][k][l]
                assert np.isfinite(got).all(), (math, l, k)
                assert np.abs(got - ref).max() / scale < 2e-6, (math, l, k)
        np.testing.assert_array_equal(out["f16x2"][0][l], out["fp32"][0][l])
