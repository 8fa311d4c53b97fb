"""Every selectable kernel path must agree with the default path (and hence with the oracle): the any-shape
fall-backs (NG_MP_PATH / NG_EDGE_PATH / NG_FC_PATH = layered, NG_DENSE_PATH / NG_HEAD_PATH = generic) and the
strict f32-input-MFMA arithmetic (NG_EDGE_MATH / NG_EDGE_BWD_MATH / NG_GEMM_MATH = fp32).  The switches are read
once per process (ng_reload_env re-reads them; tests/conftest.py's monkeypatch wrapper calls it)."""
import numpy as np
import pytest

from helpers import make_hp, small_batch, randomize_biases, rel_err

pytestmark = pytest.mark.gpu

VARIANTS = [
    {"NG_MP_PATH": "layered"}, {"NG_EDGE_PATH": "layered"}, {"NG_DENSE_PATH": "generic"}, {"NG_FC_PATH": "layered"},
    {"NG_HEAD_PATH": "generic"}, {"NG_EDGE_MATH": "fp32"}, {"NG_EDGE_BWD_MATH": "fp32"}, {"NG_GEMM_MATH": "fp32"},
    {"NG_MP_PATH": "layered", "NG_DENSE_PATH": "generic", "NG_GEMM_MATH": "fp32"},
]


def _run(gpu_device, eng, gb, xi, mask, dpe):
    pk = eng.forward(gb, training=True, noise=xi, dropout_mask=mask).clone()
    eng.backward(dpe)
    inf = eng.forward(gb).clone()
    return pk.cpu().numpy(), inf.cpu().numpy(), eng.params.grads_dict()


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_variant_matches_default(gpu_device, monkeypatch, env):
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    hp = make_hp(atom_feature_size=64, edge_feature_size=3, edge_hidden_size=128)
    b = small_batch(5, 77, seed=11)          # 385 atoms: ragged last tiles everywhere
    eng = Engine(hp, 10, device=gpu_device, seed=3)
    randomize_biases(eng)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"],
                    device=gpu_device)
    N, K = b["edges"].shape
    xi = eng.randn(N * K, seed=5)
    mask = eng.dropout_mask(N * 32, seed=6)
    dpe = torch.from_numpy(np.random.default_rng(1).standard_normal(N).astype(np.float32)).to(gpu_device)
    base = _run(gpu_device, eng, gb, xi, mask, dpe)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    var = _run(gpu_device, eng, gb, xi, mask, dpe)
    assert np.max(np.abs(var[0] - base[0])) < 5e-5
    assert np.max(np.abs(var[1] - base[1])) < 5e-5
    for k in base[2]:
        assert rel_err(var[2][k], base[2][k]) < 2e-4, k


@pytest.mark.parametrize("N,K,E,span", [(1000, 16, 3, 200), (1000, 16, 3, 0), (333, 5, 2, 100), (70, 24, 1, 0),
                                        (4096, 16, 3, 256), (31, 16, 3, 0)])
@pytest.mark.parametrize("path", ["default", "layered"])
def test_mp_layer_paths_vs_numpy(gpu_device, monkeypatch, path, N, K, E, span):
    """ng_mp_layer_fwd on the F=64 fast paths: local neighbour windows (span > 0: neighbours within
    +-span rows -> LDS window, incl. restaging as the run moves), unrestricted lists (span = 0 -> the
    global-gather branch), padded slots, ragged tails, K % 4 != 0."""
    import ctypes as C
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    if path != "default":
        monkeypatch.setenv("NG_MP_PATH", path)
    rng = np.random.default_rng(N + K)
    F = 64
    h = rng.standard_normal((N, F)).astype(np.float32)
    if span:
        base = np.arange(N)[:, None]
        nl = np.clip(base + rng.integers(-span // 2, span // 2, (N, K)), 0, N - 1).astype(np.int32)
    else:
        nl = rng.integers(0, N, (N, K)).astype(np.int32)
    e = rng.standard_normal((N, K, E)).astype(np.float32)
    pad = rng.random((N, K)) < 0.1
    e[pad] = 0.0
    nl[pad] = 0                                        # padded slots point at row 0 (far outside any window)
    inv = rng.random(N).astype(np.float32)
    w = (rng.standard_normal((F, F, E)) * 0.1).astype(np.float32)
    P = np.einsum("ijn,ijl,lmn,i->im", e.astype(np.float64), h.astype(np.float64)[nl], w.astype(np.float64),
                  inv.astype(np.float64))
    ref = np.log1p(np.exp(-np.abs(P))) + np.maximum(P, 0) + h
    refA = np.einsum("ijn,ijl->inl", e.astype(np.float64), h.astype(np.float64)[nl])
    dev = gpu_device
    th, tn, te, ti, tw = (torch.from_numpy(x).to(dev) for x in (h, nl, e, inv, w))
    out = torch.empty(N, F, device=dev)
    A = torch.empty(N, E, F, device=dev)
    S = torch.empty(N, F, device=dev)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.ng_mp_layer_fwd(ctx.handle, st, N, K, F, E, 1, 1, ptr(th), ptr(tn), ptr(te), ptr(ti),
                                      ptr(tw), ptr(out), ptr(A), ptr(S)), "mp")
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(A.cpu().numpy(), refA, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(S.cpu().numpy(), ref - h, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("fwd_env,bwd_env", [({}, {"NG_EDGE_BWD_MATH": "fp32"}), ({"NG_EDGE_BWD_MATH": "fp32"}, {}),
                                             ({}, {"NG_EDGE_MATH": "fp32"})])
def test_switch_flipped_between_forward_and_backward(gpu_device, monkeypatch, fwd_env, bwd_env):
    """the edge tape's element order is fixed when the forward writes it and handed to the backward
    (ng_edge_tape_layout -> ng_edge_mlp_bwd_tape): flipping a kernel switch in between must not change the gradients"""
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    hp = make_hp(atom_feature_size=64, edge_feature_size=3, edge_hidden_size=128)
    b = small_batch(5, 77, seed=11)
    eng = Engine(hp, 10, device=gpu_device, seed=3)
    randomize_biases(eng)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    N, K = b["edges"].shape
    xi = eng.randn(N * K, seed=5)
    mask = eng.dropout_mask(N * 32, seed=6)
    dpe = torch.from_numpy(np.random.default_rng(1).standard_normal(N).astype(np.float32)).to(gpu_device)
    eng.forward(gb, training=True, noise=xi, dropout_mask=mask)
    eng.backward(dpe)
    base = eng.params.grads_dict()
    for k, v in fwd_env.items():
        monkeypatch.setenv(k, v)
    eng.forward(gb, training=True, noise=xi, dropout_mask=mask)
    for k in fwd_env:
        monkeypatch.delenv(k)
    for k, v in bwd_env.items():
        monkeypatch.setenv(k, v)
    eng.backward(dpe)
    var = eng.params.grads_dict()
    for k in base:
        assert rel_err(var[k], base[k]) < 2e-4, k


@pytest.mark.parametrize("deferred", [True, False])
def test_wide_second_stage_reduction_gives_the_narrow_form_s_bits(gpu_device, monkeypatch, deferred):
    """reduce.cuh: jobs of 32 K elements and more run 256 elements per block, four per lane; per element the order of the sum is
    that of the 64-element form (NG_REDUCE=narrow), so every weight gradient of a default-width backward — 196,608-element MPLayer
    sums, 65,536-element dense ones, the small ones that stay narrow — must come out bit for bit, queued or eager."""
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    hp = make_hp(atom_feature_size=256, edge_feature_size=3, edge_hidden_size=128)
    b = small_batch(40, 120, seed=4)         # 4800 atoms: the generic GEMM path with split-K partials
    eng = Engine(hp, 10, device=gpu_device, seed=3)
    eng.defer_reductions = deferred
    randomize_biases(eng)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    N, K = b["edges"].shape
    xi = eng.randn(N * K, seed=5)
    mask = eng.dropout_mask(N * 128, seed=6)
    dpe = torch.from_numpy(np.random.default_rng(1).standard_normal(N).astype(np.float32)).to(gpu_device)
    wide = _run(gpu_device, eng, gb, xi, mask, dpe)
    monkeypatch.setenv("NG_REDUCE", "narrow")
    narrow = _run(gpu_device, eng, gb, xi, mask, dpe)
    assert np.array_equal(wide[0], narrow[0])
    for k in wide[2]:
        assert np.array_equal(wide[2][k], narrow[2][k]), k
