"""Live-edge view of the padded lists (round 4; include/nmrgnn_hip.h: ng_build_live_edges, ng_add_noise_live,
ng_edge_mlp_fwd_live / _bwd_live).  A padded slot (edges == 0) yields e == 0 and no gradient contribution
(nmrgnn/model.py:251,257,261), so the fused edge kernels may skip it: e must come out BIT FOR BIT as from the
every-slot kernels, the tape rows of the live slots too, and the weight gradients up to the order of summation
over edges."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_hp, randomize_biases, rel_err, small_batch
from test_gpu_edge_h2 import H, tape_perm

pytestmark = pytest.mark.gpu


def _ctx():
    from nmrgnn_amd import _lib
    return _lib.get_context(0)


def _st(dev):
    import torch
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def build_live(dev, edges):
    import torch
    from nmrgnn_amd._lib import ptr
    n = edges.numel()
    perm = torch.full((max(n, 1),), -7, dtype=torch.int32, device=dev)
    pos = torch.full((max(n, 1),), -7, dtype=torch.int32, device=dev)
    d_c = torch.full((max(n, 1),), -7.0, dtype=torch.float32, device=dev)
    n_live = torch.full((1,), -7, dtype=torch.int32, device=dev)
    ctx = _ctx()
    ctx.check(ctx.lib.ng_build_live_edges(ctx.handle, _st(dev), n, ptr(edges), ptr(perm), ptr(pos), ptr(d_c), ptr(n_live)),
              "ng_build_live_edges")
    return perm, pos, d_c, n_live


@pytest.mark.parametrize("n,p_dead", [(1, 0.0), (1, 1.0), (1023, 0.3), (1024, 0.05), (1025, 0.5), (70001, 0.1), (4096, 1.0),
                                      (5000, 0.0)])
def test_live_partition_is_the_stable_one(gpu_device, n, p_dead):
    import torch
    rng = np.random.default_rng(n)
    d = rng.uniform(0.05, 1.0, n).astype(np.float32)
    d[rng.random(n) < p_dead] = 0.0
    perm, pos, d_c, n_live = build_live(gpu_device, torch.from_numpy(d).to(gpu_device))
    live = np.nonzero(d > 0)[0]
    dead = np.nonzero(~(d > 0))[0]
    assert int(n_live.cpu()) == len(live)
    assert np.array_equal(perm.cpu().numpy()[:n], np.concatenate([live, dead]))
    want_pos = np.full(n, -1, np.int64)
    want_pos[live] = np.arange(len(live))
    assert np.array_equal(pos.cpu().numpy()[:n], want_pos)
    assert np.array_equal(d_c.cpu().numpy()[:len(live)], d[live])


def test_noise_in_the_compacted_order_carries_the_bits_of_add_noise(gpu_device):
    import torch
    from nmrgnn_amd._lib import ptr
    n = 50001
    rng = np.random.default_rng(3)
    d = rng.uniform(0.05, 1.0, n).astype(np.float32)
    d[rng.random(n) < 0.2] = 0.0
    td = torch.from_numpy(d).to(gpu_device)
    perm, pos, d_c, n_live = build_live(gpu_device, td)
    ctx = _ctx()
    full = torch.empty(n, device=gpu_device)
    ctx.check(ctx.lib.ng_add_noise(ctx.handle, _st(gpu_device), 11, 5, n, ptr(td), 0.025, ptr(full)), "ng_add_noise")
    comp = torch.full((n,), -7.0, device=gpu_device)
    ctx.check(ctx.lib.ng_add_noise_live(ctx.handle, _st(gpu_device), 11, 5, n, ptr(td), None, 0.025, ptr(pos), ptr(comp)),
              "ng_add_noise_live")
    nl = int(n_live.cpu())
    live = np.nonzero(d > 0)[0]
    assert np.array_equal(comp.cpu().numpy()[:nl], full.cpu().numpy()[live])
    assert np.all(comp.cpu().numpy()[nl:] == -7.0)
    # explicit draws
    xi = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(gpu_device)
    ctx.check(ctx.lib.ng_add_scaled(ctx.handle, _st(gpu_device), n, ptr(td), ptr(xi), 0.025, ptr(full)), "ng_add_scaled")
    ctx.check(ctx.lib.ng_add_noise_live(ctx.handle, _st(gpu_device), 0, 0, n, ptr(td), ptr(xi), 0.025, ptr(pos), ptr(comp)),
              "ng_add_noise_live")
    assert np.array_equal(comp.cpu().numpy()[:nl], full.cpu().numpy()[live])


def _weights(rng, E):
    Ws = [rng.standard_normal((H, H)) * 0.12 for _ in range(3)] + [rng.standard_normal((H, E)) * 0.1]
    bs = [rng.standard_normal(H) * 0.1 for _ in range(3)] + [rng.standard_normal(E) * 0.1]
    return Ws, bs


def _edge_pair(dev, d_src, d_eff, E, Ws, bs, de, live):
    """forward (+ tape) and backward of the edge path; live = None: every slot; else the live-view entry points"""
    import torch
    from nmrgnn_amd._lib import ptr, ptr_array
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    n = len(d_src)
    centers = np.linspace(0.0, 1.2, H)
    gap = float(centers[1] - centers[0])
    tc = t(centers)
    tW, tb = [t(w) for w in Ws], [t(b) for b in bs]
    e = torch.full((n, E), 7.0, device=dev)
    z = torch.full((3, n, H), 7.0, device=dev)
    dW = [torch.full_like(w, 7.0) for w in tW]
    db = [torch.full_like(b, 7.0) for b in tb]
    tde = t(de)
    ctx = _ctx()
    st = _st(dev)
    layout = int(ctx.lib.ng_edge_tape_layout(H, E, 4, 1, n))
    if live is None:
        td, te = t(d_src), t(d_eff)
        ctx.check(ctx.lib.ng_edge_mlp_fwd(ctx.handle, st, n, H, E, 4, 1, ptr(td), ptr(te), ptr(tc), gap, ptr_array(tW),
                                          ptr_array(tb), ptr(e), ptr(z)), "fwd")
        ctx.check(ctx.lib.ng_edge_mlp_bwd_tape(ctx.handle, st, n, H, E, 4, 1, ptr(td), ptr(te), ptr(tc), gap, ptr_array(tW),
                                               ptr(z), ptr(tde), ptr_array(dW), ptr_array(db), layout), "bwd")
    else:
        perm, pos, d_c, n_live = live
        idx = perm.cpu().numpy()[:int(n_live.cpu())]
        te = torch.full((n,), 3.0, device=dev)
        te[:len(idx)] = t(d_eff[idx])
        ctx.check(ctx.lib.ng_edge_mlp_fwd_live(ctx.handle, st, n, H, E, 4, 1, ptr(d_c), ptr(te), ptr(perm), ptr(n_live),
                                               ptr(tc), gap, ptr_array(tW), ptr_array(tb), ptr(e), ptr(z)), "fwd_live")
        ctx.check(ctx.lib.ng_edge_mlp_bwd_live(ctx.handle, st, n, H, E, 4, 1, ptr(d_c), ptr(te), ptr(perm), ptr(n_live),
                                               ptr(tc), gap, ptr_array(tW), ptr(z), ptr(tde), ptr_array(dW), ptr_array(db),
                                               layout), "bwd_live")
    torch.cuda.synchronize()
    return (e.cpu().numpy(), z.cpu().numpy(), [w.cpu().numpy() for w in dW], [b.cpu().numpy() for b in db], layout)


def _rows(z_layer, n_rows, layout):
    """row-major [n_rows][H] view of the first n_rows rows of one tape layer"""
    flat = z_layer.reshape(-1)[:n_rows * H]
    if layout:
        flat = flat[tape_perm(n_rows)]
    return flat.reshape(n_rows, H)


@pytest.mark.parametrize("n,E,p_dead,math", [(70001, 3, 0.1, ""), (4096, 3, 0.5, ""), (777, 2, 0.9, ""), (513, 3, 0.0, ""),
                                             (300, 3, 1.0, ""), (20000, 3, 0.2, "fp32"), (64, 1, 0.3, "")])
def test_edge_pair_over_the_live_view_equals_every_slot(gpu_device, monkeypatch, n, E, p_dead, math):
    import torch
    if math:
        monkeypatch.setenv("NG_EDGE_MATH", math)
    rng = np.random.default_rng(n + E)
    d_src = rng.uniform(0.05, 1.2, n).astype(np.float32)
    d_src[rng.random(n) < p_dead] = 0.0
    d_eff = np.where(d_src > 0, d_src + np.float32(0.025) * rng.standard_normal(n).astype(np.float32), d_src).astype(np.float32)
    Ws, bs = _weights(rng, E)
    de = rng.standard_normal((n, E)).astype(np.float32)
    live = build_live(gpu_device, torch.from_numpy(d_src).to(gpu_device))
    e0, z0, dW0, db0, lay = _edge_pair(gpu_device, d_src, d_eff, E, Ws, bs, de, None)
    e1, z1, dW1, db1, _ = _edge_pair(gpu_device, d_src, d_eff, E, Ws, bs, de, live)
    assert np.array_equal(e0, e1)                       # dead slots exactly 0, live slots the same bits
    idx = np.nonzero(d_src > 0)[0]
    assert np.all(e1[d_src <= 0] == 0.0)
    nl = len(idx)
    if nl:
        for l in range(3):
            assert np.array_equal(_rows(z1[l], nl, lay), _rows(z0[l], n, lay)[idx])
    for a, b in zip(dW0 + db0, dW1 + db1):
        scale = np.abs(a).max() + 1e-30
        assert np.abs(a - b).max() <= 2e-6 * scale + 1e-12, (np.abs(a - b).max(), scale)
    if nl == 0:
        assert all(np.all(w == 0) for w in dW1 + db1)


@pytest.mark.parametrize("F", [64, 256])
def test_engine_with_and_without_the_live_view(gpu_device, F):
    """whole model: peaks bit for bit, every gradient to summation order; one member graph smaller than K (most of its
    slots padded) and a graph of a single atom (no live slot at all)"""
    import torch
    from nmrgnn_amd import synth
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch, concat_graphs
    rng = np.random.default_rng(5)
    graphs = []
    for n_at in (40, 9, 1, 33):
        a, nl, d = synth.make_graph(n_at, 16, 10, 0.1, rng)
        graphs.append((a, nl, d, synth.inv_degree(nl)))
    gb = concat_graphs(graphs, device=gpu_device)
    hp = make_hp(atom_feature_size=F)
    eng = Engine(hp, 10, device=gpu_device, seed=3)
    randomize_biases(eng)
    N, K = gb.edges.shape
    xi = eng.randn(N * K, seed=1)
    mask = eng.dropout_mask(N * (F // 2), seed=2)
    dp = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(gpu_device)
    out = {}
    for live in (False, True):
        eng.use_live_edges = live
        inf = eng.forward(gb).cpu().numpy()
        peaks = eng.forward(gb, training=True, noise=xi, dropout_mask=mask)
        assert (eng.tape.live is not None) == live
        eng.backward(dp)
        torch.cuda.synchronize()
        out[live] = (inf, peaks.cpu().numpy(), {k: v.copy() for k, v in eng.params.grads_dict().items()})
        # the GPU draw of the noise goes through the compacted order as well
        drawn = eng.forward(gb, training=True, seed=9, dropout_mask=mask).cpu().numpy()
        out[live] += (drawn,)
    assert np.array_equal(out[False][0], out[True][0])
    assert np.array_equal(out[False][1], out[True][1])
    assert np.array_equal(out[False][3], out[True][3])
    for k, v in out[False][2].items():
        assert rel_err(out[True][2][k], v) < 5e-6, k
