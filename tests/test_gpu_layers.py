"""Layer classes of the drop-in surface on the GPU: the reference's own shape tests
(reference tests/test_nmrgnn.py:18-108) plus the analytic KATs of SURVEY App. C through the HIP path."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_hp

pytestmark = pytest.mark.gpu


def ring(F=16, E=2):
    nodes = np.eye(F, dtype=np.float32)[[2, 4, 0, 1, 3]]
    nlist = np.zeros((5, 2), dtype=np.int64)
    for i in range(5):
        for k, j in enumerate(range(-1, 3, 2)):
            nlist[i, k] = (i + j) % 5
    return nodes, nlist, np.ones((5, 2, E), np.float32), (np.ones(5) / 2)


def test_mplayer_kats(gpu_device):
    import torch
    import nmrgnn_amd
    nodes, nlist, edges, inv = ring()
    mpl = nmrgnn_amd.MPLayer()                      # activation=None, like the reference default
    out = mpl([nodes, nlist, edges, inv])
    assert tuple(out.shape) == nodes.shape          # reference assertion (tests:34)
    mpl.w = torch.ones(16, 16, 2, device=gpu_device)
    np.testing.assert_allclose(mpl([nodes, nlist, edges, inv]).cpu().numpy(), 2.0, rtol=1e-6)   # KAT-1
    l, m, n = np.meshgrid(np.arange(16), np.arange(16), np.arange(2), indexing="ij")
    mpl.w = torch.tensor(((l + 1) * (m + 1) * (n + 1) / 100.0).astype(np.float32), device=gpu_device)
    out = mpl([nodes, nlist, edges, inv]).cpu().numpy()
    np.testing.assert_allclose(out[0], 0.135 * (np.arange(16) + 1), rtol=1e-5)                  # KAT-2


def test_rbf_kat5(gpu_device):
    import nmrgnn_amd
    from oracle import nmrgnn_oracle as O
    rbf = nmrgnn_amd.RBFExpansion(0.005, 0.20, 128)
    centers, gap = O.rbf_centers(0.005, 0.20, 128)
    d = np.array([[centers[17], centers[40] + np.sqrt(gap)]], np.float32)
    out = rbf(d).cpu().numpy()
    assert out.shape == (1, 2, 128)
    assert out[0, 0, 17] == pytest.approx(1.0, abs=1e-6)
    assert out[0, 1, 40] == pytest.approx(np.exp(-1.0), rel=1e-4)
    np.testing.assert_allclose(out, O.rbf_expand(d, centers, gap), atol=2e-6)


def test_block_shapes_like_reference_tests(gpu_device):
    import nmrgnn_amd
    hp = nmrgnn_amd.build_GNNModel().hypers
    e = nmrgnn_amd.EdgeFCBlock(hp)(np.ones((5, 2, 2), np.float32))            # tests:66-73
    assert e.shape[-1] == hp.get('edge_feature_size') and tuple(e.shape[:-1]) == (5, 2)
    nodes, nlist, edges, inv = ring(F=16, E=2)
    out = nmrgnn_amd.MPBlock(hp)([nodes, nlist, edges, inv])                   # tests:78-96
    assert tuple(out.shape) == nodes.shape
    x = np.ones((5, hp.get('atom_feature_size')), np.float32)
    y = nmrgnn_amd.FCBlock(hp)(x)                                              # tests:101-108
    assert y.shape[-1] == hp.get('atom_feature_size') // 2


def test_gnnmodel_call_like_reference_test(gpu_device):
    import nmrgnn_amd
    nodes = np.eye(16, dtype=np.float32)[[2, 4, 1, 3, 3]]                       # tests:198
    _, nlist, _, inv = ring()
    model = nmrgnn_amd.build_GNNModel()
    out = model([nodes, nlist, np.ones((5, 2), np.float32), inv])
    assert out.shape == (5,)
    assert model([nodes, nlist, np.ones((5, 2), np.float32), inv]).shape == (5,)   # tests:222-223


@pytest.mark.parametrize("F,E", [(64, 3), (32, 2), (256, 3), (128, 8)])
def test_aggregate_abi_matches_numpy(gpu_device, F, E):
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    rng = np.random.default_rng(0)
    N, K = 301, 16
    h = rng.standard_normal((N, F)).astype(np.float32)
    nl = rng.integers(0, N, (N, K)).astype(np.int32)
    e = rng.standard_normal((N, K, E)).astype(np.float32)
    ref = np.einsum("ijn,ijl->inl", e.astype(np.float64), h.astype(np.float64)[nl])
    ctx = _lib.get_context(0)
    dev = gpu_device
    th, tn, te = (torch.from_numpy(x).to(dev) for x in (h, nl, e))
    A = torch.empty(N, E, F, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.ng_mp_aggregate(ctx.handle, st, N, K, F, E, ptr(th), ptr(tn), ptr(te), ptr(A)), "agg")
    np.testing.assert_allclose(A.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("F,E,act", [(16, 2, None), (64, 3, "softplus"), (256, 8, "tanh")])
def test_amplayer_matches_oracle(gpu_device, F, E, act):
    """AMPLayer (nmrgnn/layers.py:48-100): reference shape test (tests:46-52) + literal fp64 oracle."""
    import nmrgnn_amd
    from oracle import nmrgnn_oracle as O
    rng = np.random.default_rng(3)
    N, K = 203, 16
    nodes = rng.standard_normal((N, F)).astype(np.float32)
    nlist = rng.integers(0, N, (N, K))
    edges = rng.standard_normal((N, K, E)).astype(np.float32)
    edges[:, K - 3:, :] = 0.0                                  # padded slots still take part in the softmax
    nlist[:, K - 3:] = 0
    inv = (1.0 / 13) * np.ones(N)
    layer = nmrgnn_amd.AMPLayer(activation=act)
    out = layer([nodes, nlist, edges, inv])
    assert tuple(out.shape) == nodes.shape
    ref = O.amp_layer_forward(nodes, nlist, edges, inv, layer.wq.cpu().numpy(), layer.wk.cpu().numpy(),
                              layer.wv.cpu().numpy(), act)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("F,E,act", [(16, 2, None), (64, 3, "softplus"), (256, 8, "tanh"), (20, 5, "relu")])
def test_amplayer_backward_matches_oracle(gpu_device, F, E, act):
    """Gradients of AMPLayer (layers.py:48-100) w.r.t. nodes, edges, wq, wk, wv vs the fp64 reverse pass of the oracle;
    padded slots (nlist 0, edges 0) take part in the softmax and receive gradient, as in the reference.  Run twice: the
    backward has no atomics, so the results are bit-identical."""
    import torch
    import nmrgnn_amd
    from oracle import nmrgnn_oracle as O
    rng = np.random.default_rng(5)
    N, K = 301, 16
    nodes = rng.standard_normal((N, F)).astype(np.float32)
    nlist = rng.integers(0, N, (N, K))
    edges = rng.standard_normal((N, K, E)).astype(np.float32)
    edges[:, K - 3:, :] = 0.0
    nlist[:, K - 3:] = 0
    inv = (1.0 / rng.integers(1, K, N)).astype(np.float32)
    dout = rng.standard_normal((N, F)).astype(np.float32)
    layer = nmrgnn_amd.AMPLayer(activation=act)
    layer([nodes, nlist, edges, inv])
    dn, de = layer.backward(dout)
    got = dict(nodes=dn, edges=de, **layer.grads)
    ref = O.amp_layer_backward(nodes, nlist, edges, inv, layer.wq.cpu().numpy(), layer.wk.cpu().numpy(),
                               layer.wv.cpu().numpy(), act, dout)
    for name, r in ref.items():
        g = got[name].cpu().numpy()
        scale = np.abs(r).max()
        assert np.abs(g - r).max() <= 2e-5 * max(1.0, scale), (name, np.abs(g - r).max(), scale)
    dn2, de2 = layer.backward(dout)
    assert torch.equal(dn, dn2) and torch.equal(de, de2)
    for name in ("wq", "wk", "wv"):
        assert torch.equal(got[name], layer.grads[name])


def test_mplayer_kernel_regularizer_adds_a_loss(gpu_device):
    """layers.py:9,44-45: MPLayer(kernel_regularizer=...) adds regularizer(w) to the layer's losses on every call
    (keras identifiers 'l1' / 'l2' default to factor 0.01; callables are applied as given; junk raises)."""
    import nmrgnn_amd
    nodes = np.eye(16, dtype=np.float32)[[2, 4, 0, 1, 3]]
    nlist = np.array([[(i - 1) % 5, (i + 1) % 5] for i in range(5)], np.int32)
    edges = np.ones((5, 2, 2), np.float32)
    inv = np.full(5, 0.5, np.float32)
    plain = nmrgnn_amd.MPLayer()
    plain([nodes, nlist, edges, inv])
    assert plain.losses == []
    l2 = nmrgnn_amd.MPLayer(kernel_regularizer='l2')
    out = l2([nodes, nlist, edges, inv])
    assert out.shape == (5, 16) and len(l2.losses) == 1
    w = l2.w.cpu().numpy().astype(np.float64)
    assert float(l2.losses[0]) == pytest.approx(0.01 * (w ** 2).sum(), rel=1e-5)
    custom = nmrgnn_amd.MPLayer(kernel_regularizer=lambda w: 3.0 * w.abs().max())
    custom([nodes, nlist, edges, inv])
    assert float(custom.losses[0]) == pytest.approx(3.0 * np.abs(custom.w.cpu().numpy()).max(), rel=1e-6)
    with pytest.raises(ValueError):
        nmrgnn_amd.MPLayer(kernel_regularizer='no-such-regularizer')


@pytest.mark.parametrize("F,E,K,graph", [(256, 3, 16, 256), (128, 2, 8, 200), (256, 1, 16, 300)])
def test_slab_window_aggregation_equals_the_gather_kernel(gpu_device, F, E, K, graph):
    """At F % 128 == 0 and a batch of small graphs (ng_ctx_set_graph_span <= 272) ng_mp_aggregate keeps 128-column slab
    windows of h in LDS (mp_win.hip: agg_win_kernel); it sums a row's neighbours in entry order with fused multiply-adds
    like the L2-gather kernel, so the two agree BIT FOR BIT — also for graphs that straddle tiles (200, 300 atoms: tiles
    whose range leaves the window restage it or gather from global memory) and for a ragged last tile."""
    import ctypes as C
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    rng = np.random.default_rng(F + K)
    G = 24
    N = G * graph - 7                                   # ragged end
    base = (np.arange(N) // graph) * graph
    nl = np.minimum(base[:, None] + rng.integers(0, graph, (N, K)), N - 1).astype(np.int32)
    e = rng.standard_normal((N, K, E)).astype(np.float32)
    e[rng.random((N, K)) < 0.1] = 0.0
    h = rng.standard_normal((N, F)).astype(np.float32)
    th, tn, te = (torch.from_numpy(a).to(gpu_device) for a in (h, nl, e))
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
    out = {}
    for span in (0, graph):
        ctx.check(ctx.lib.ng_ctx_set_graph_span(ctx.handle, span), "span")
        A = torch.full((N, E, F), 7.0, device=gpu_device)
        ctx.check(ctx.lib.ng_mp_aggregate(ctx.handle, st, N, K, F, E, ptr(th), ptr(tn), ptr(te), ptr(A)), "agg")
        torch.cuda.synchronize()
        out[span] = A
    ctx.check(ctx.lib.ng_ctx_set_graph_span(ctx.handle, 0), "span")
    ref = np.einsum("ijn,ijl->inl", e.astype(np.float64), h.astype(np.float64)[nl])
    assert np.abs(out[0].cpu().numpy() - ref).max() < 1e-4
    if graph <= 272:
        assert torch.equal(out[0], out[graph])
    else:
        assert torch.equal(out[0], out[graph])          # span above the window: the hint selects the gather kernel
