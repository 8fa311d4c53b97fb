"""Host-side logic that needs no GPU: hyper-parameters, parameter layout, synthetic graphs,
graph batching / transposed edge lists, sharding."""
import numpy as np
import pytest
import torch

from nmrgnn_amd import synth
from nmrgnn_amd.graph import GraphBatch, concat_graphs
from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
from nmrgnn_amd.parallel import shard_range
from nmrgnn_amd.params import param_shapes


def test_hyper_defaults_follow_reference():
    hp = declare_gnn_space(HyperParameters())          # nmrgnn/model.py:22-36,45
    assert hp.get('atom_feature_size') == 256 and hp.get('edge_feature_size') == 3
    assert hp.get('edge_hidden_size') == 128 and hp.get('mp_layers') == 4
    assert hp.get('fc_layers') == 4 and hp.get('edge_fc_layers') == 4
    assert hp.get('noise') == 0.025 and hp.get('dropout') is True
    assert hp.get('rbf_low') == 0.005 and hp.get('rbf_high') == 0.20
    assert hp.get('learning_rate') == 1e-4
    with pytest.raises(ValueError):
        declare_gnn_space(HyperParameters(atom_feature_size=100))
    with pytest.raises(KeyError):
        hp.get('nope')


def test_parameter_count_matches_bundled_model():
    hp = declare_gnn_space(HyperParameters())
    n = sum(int(np.prod(s)) for _, s in param_shapes(hp, 10))
    assert n == 1_070_477
    hp = declare_gnn_space(HyperParameters(atom_feature_size=64))
    assert sum(int(np.prod(s)) for _, s in param_shapes(hp, 10)) == 114_605


def test_synthetic_batch_shape_and_conventions():
    b = synth.make_batch(4, 256, 16, 10, 0.05, seed=1)
    N = 4 * 256
    assert b["atoms"].shape == (N, 10) and b["nlist"].shape == (N, 16)
    assert np.all(b["atoms"].sum(1) == 1) and set(np.argmax(b["atoms"], 1)) <= {2, 3, 4}
    pad = b["edges"] == 0
    assert 0.02 < pad.mean() < 0.09
    real = ~pad
    assert b["edges"][real].min() >= 0.09 and b["edges"][real].max() <= 0.45
    for g in range(4):
        sl = slice(g * 256, (g + 1) * 256)
        nl = b["nlist"][sl]
        assert nl.min() >= g * 256 and nl.max() < (g + 1) * 256
        loc = nl - g * 256
        rows = np.arange(256)[:, None]
        assert not np.any((loc == rows) & real[sl])                    # no self edges
        for i in range(0, 256, 37):
            r = loc[i][real[sl][i]]
            assert len(set(r)) == len(r)                               # distinct neighbours
        deg = (loc > 0).sum(1)
        np.testing.assert_allclose(b["inv_degree"][sl], np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0))
    assert np.any((b["nlist"][:256] == 0) & real[:256])              # a real edge to atom 0


def test_csc_lists_cover_exactly_the_unmasked_edges():
    b = synth.make_batch(2, 30, 8, 10, 0.2, seed=3)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"],
                    device="cpu")
    ptr, eid = gb.csc()
    ptr, eid = ptr.numpy(), eid.numpy()
    N, K = b["nlist"].shape
    flat_nl, flat_e = b["nlist"].reshape(-1), b["edges"].reshape(-1)
    assert ptr[0] == 0 and ptr[-1] == (flat_e > 0).sum() == len(eid)
    for t in range(N):
        mine = eid[ptr[t]:ptr[t + 1]]
        assert np.all(flat_nl[mine] == t) and np.all(flat_e[mine] > 0)
        assert np.all(np.diff(mine) > 0)                                # deterministic order
    assert len(set(eid.tolist())) == len(eid)


def test_concat_graphs_offsets_indices():
    g1 = synth.make_graph(10, 4, 10, 0.0, np.random.default_rng(0))
    g2 = synth.make_graph(7, 4, 10, 0.0, np.random.default_rng(1))
    t1 = (g1[0], g1[1], g1[2], synth.inv_degree(g1[1]))
    t2 = (g2[0], g2[1], g2[2], synth.inv_degree(g2[1]))
    gb = concat_graphs([t1, t2], device="cpu")
    assert gb.N == 17 and gb.G == 2 and list(gb.graph_ptr_host) == [0, 10, 17]
    np.testing.assert_array_equal(gb.nlist[10:].numpy(), g2[1] + 10)
    with pytest.raises(ValueError):
        GraphBatch(g1[0], g1[1] + 100, g1[2], t1[3], device="cpu")


def test_shard_range_partitions_everything():
    for n, w in [(4096, 8), (10, 3), (5, 8), (0, 2)]:
        seen = []
        for r in range(w):
            lo, hi = shard_range(n, r, w)
            seen += list(range(lo, hi))
        assert seen == list(range(n))
    assert shard_range(4096, 3, 8) == (1536, 2048)


def test_two_piece_fp16_split_bounds():
    """host restatement of the split used on the device (csrc/h2_common.cuh): h = rne_f16(x), l = rne_f16(x - h) with
    the subtraction exact in fp32; x - h - l is at most max(2^-22 |x|, 2^-25) (l may be an fp16 subnormal, which the
    matrix pipe honours), and a product taken as lh + hl + hh misses a*b by no more than the two representation errors
    plus the dropped l*l <= 2^-22 |a b|"""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(8192), rng.standard_normal(4096) * 1e-3, rng.uniform(-6e4, 6e4, 4096)]
                       ).astype(np.float32)
    h = x.astype(np.float16)
    r = x - h.astype(np.float32)
    assert np.all(r.astype(np.float64) == x.astype(np.float64) - h.astype(np.float64))      # exact in fp32
    l = r.astype(np.float16)
    rest = x.astype(np.float64) - h.astype(np.float64) - l.astype(np.float64)
    assert np.all(np.abs(rest) <= np.maximum(2.0 ** -22 * np.abs(x.astype(np.float64)), 2.0 ** -25))
    # typical (rms) representation error: ~2^-23 |x|, twice the unit of fp32's own rounding
    big = np.abs(x) > 1e-2
    assert np.sqrt(np.mean((rest[big] / x[big]) ** 2)) < 2.0 ** -22.5
    # three piece products against the exact product
    y = rng.standard_normal(x.size).astype(np.float32)
    hy = y.astype(np.float16)
    ly = (y - hy.astype(np.float32)).astype(np.float16)
    f = lambda a: a.astype(np.float64)
    prod3 = f(l) * f(hy) + f(h) * f(ly) + f(h) * f(hy)
    exact = f(x) * f(y)
    ax, ay = np.abs(f(x)), np.abs(f(y))
    bound = ax * np.maximum(2.0 ** -22 * ay, 2.0 ** -25) + ay * np.maximum(2.0 ** -22 * ax, 2.0 ** -25) + 2.0 ** -21.9 * ax * ay
    assert np.all(np.abs(prod3 - exact) <= bound)


def test_batch_prefetcher_on_a_host_device_builds_each_batch_when_asked():
    """without a GPU the prefetcher has no side stream: it hands out plain GraphBatches, in order, with graph_ptr kept"""
    import torch
    from nmrgnn_amd import synth
    from nmrgnn_amd.graph import BatchPrefetcher, GraphBatch
    bs = [synth.make_batch(2, 20 + i, 16, 10, 0.1, seed=i) for i in range(3)]
    items = [((b["atoms"], b["nlist"], b["edges"], b["inv_degree"]), b["graph_ptr"]) for b in bs]
    got = list(BatchPrefetcher(items, device="cpu"))
    assert [g.N for g in got] == [40, 42, 44]
    for b, g in zip(bs, got):
        ref = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device="cpu")
        assert g.G == 2 and torch.equal(g.csc()[0], ref.csc()[0]) and torch.equal(g.csc()[1], ref.csc()[1])
    assert list(BatchPrefetcher([], device="cpu")) == []
