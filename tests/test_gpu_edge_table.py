"""Edge path through a table of the edge function (csrc/edge_table.hip, Engine.edge_table — the default on batches of 256 K
edges and more since round 6, behind a device-side guard): e, peaks and every gradient against the per-edge kernels and the
float64 oracle; run-to-run bits; the guard (forced, and tripped by weights that make the edge function too sharp for the
table); the table kept over calls under frozen weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engines(F, dev, seed=3, E=3):
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
    hp = declare_gnn_space(HyperParameters(atom_feature_size=F, edge_feature_size=E, edge_hidden_size=128, mp_layers=4,
                                           fc_layers=4, edge_fc_layers=4))
    a, b = Engine(hp, 10, device=dev, seed=seed), Engine(hp, 10, device=dev, seed=seed)
    a.edge_table = False
    b.edge_table = True
    b.edge_table_min_edges = 0
    return a, b


# (E = 8, model.py:23's largest table-sized choice: 2048 points instead of 4096; the table's backward runs on the f32-input kernels)
@pytest.mark.parametrize("F,graphs,E", [(64, 24, 3), (256, 6, 3), (64, 12, 8), (64, 12, 1)])
def test_table_path_matches_the_per_edge_path(gpu_device, F, graphs, E):
    from nmrgnn_amd import synth
    from nmrgnn_amd.graph import GraphBatch
    b = synth.make_batch(graphs, 256, 16, 10, 0.05, seed=11)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    y = torch.from_numpy(b["y"]).to(gpu_device); w = torch.from_numpy(b["w"]).to(gpu_device)
    ea, eb = _engines(F, gpu_device, E=E)
    for training in (False, True):
        pa = ea.forward(gb, training=training, seed=99)
        pb = eb.forward(gb, training=training, seed=99)
        scale = float(pa.abs().max())
        assert float((pa - pb).abs().max()) <= 2e-6 * max(scale, 1.0), (training, float((pa - pb).abs().max()), scale)
    # edge features themselves: interpolation error far below fp32 resolution of the values
    assert eb.tape.table is not None and not eb.edge_table_report()[0]
    assert float((ea.tape.e - eb.tape.e).abs().max()) <= 1e-6 * max(float(ea.tape.e.abs().max()), 1.0)
    la, da = ea.loss_l2(gb, y, w, pa); lb, db = eb.loss_l2(gb, y, w, pb)
    ea.backward(da); eb.backward(db)
    ga, gbb = ea.params.grad, eb.params.grad
    for name in ea.params.offsets:
        x, z = ea.params.g(name), eb.params.g(name)
        tol = 2e-5 * max(float(x.abs().max()), 1e-12)
        assert float((x - z).abs().max()) <= tol, (name, float((x - z).abs().max()), float(x.abs().max()))
    # run to run: the fixed-point scatter does not depend on the order the edges arrive in
    pb2 = eb.forward(gb, training=True, seed=99)
    lb2, db2 = eb.loss_l2(gb, y, w, pb2)
    g1 = eb.params.grad.clone()
    eb.backward(db2)
    assert torch.equal(pb, pb2) and torch.equal(g1, eb.params.grad)


def _table_eligible(cfg):
    """the shapes the fused live-edge kernels (and with them the table path) take: E <= 8 through the fused edge MLP"""
    return cfg.get("edge_feature_size", 3) <= 8 and cfg.get("edge_hidden_size", 128) == 128 and cfg.get("edge_fc_layers", 4) == 4 \
        and cfg.get("fc_activation", "softplus") == "softplus"


def _eligible_configs():
    import test_gpu_parity as TP
    return [i for i, c in enumerate(TP.CONFIGS) if _table_eligible(c)]


@pytest.mark.parametrize("ci", _eligible_configs())
def test_table_path_against_the_oracle(gpu_device, ci):
    """the tolerances the per-edge path is held to (tests/test_gpu_parity.py): 1e-4 on shifts, 2e-4 of the largest entry on every
    gradient tensor, against the float64 oracle — training mode with explicit noise and dropout draws; every architecture of
    test_gpu_parity.CONFIGS the table path takes"""
    from oracle import nmrgnn_oracle as O
    from helpers import hp_to_oracle, rel_err
    import test_gpu_parity as TP
    hp, b, eng, sd, gb, std, avg = TP._setup(gpu_device, TP.CONFIGS[ci])
    eng.edge_table = True
    eng.edge_table_min_edges = 0
    N, K = b["edges"].shape
    Fh = hp.get('atom_feature_size') // 2
    xi = eng.randn(N * K, seed=123)
    mask = eng.dropout_mask(N * Fh, seed=321)
    peaks = eng.forward(gb, training=True, noise=xi, dropout_mask=mask)
    assert eng.tape.table is not None and not eng.edge_table_report()[0]
    dpeaks = np.random.default_rng(9).standard_normal(N).astype(np.float32)
    eng.backward(torch.from_numpy(dpeaks).to(gpu_device))
    grads = eng.params.grads_dict()
    ref_peaks, ref_grads = O.gnn_forward_backward(
        (b["atoms"], b["nlist"], b["edges"], b["inv_degree"]), sd, hp_to_oracle(hp), dpeaks, std, avg,
        training=True, noise=xi.cpu().numpy().reshape(N, K), dropout_mask=(mask.cpu().numpy().reshape(N, Fh) > 0).astype(np.float64))
    assert np.max(np.abs(peaks.cpu().numpy() - ref_peaks)) < TP.PEAK_ATOL
    bad = {k: rel_err(grads[k], g) for k, g in ref_grads.items() if rel_err(grads[k], g) > TP.GRAD_RTOL}
    assert not bad, bad


def _step(eng, gb, y, w, seed=99):
    p = eng.forward(gb, training=True, seed=seed)
    e, eng.last_table = eng.tape.e.clone(), eng.tape.table
    loss, d = eng.loss_l2(gb, y, w, p)
    eng.backward(d)
    return p.clone(), e, eng.params.grad.clone()


def test_forced_guard_gives_the_per_edge_path_bit_for_bit(gpu_device):
    """with the guard up the call IS the per-edge call: same kernels over the same rows (the table's own launches do nothing),
    e, peaks and every gradient bit for bit — also for lists without a dead slot (identity live view) and inference"""
    from nmrgnn_amd import synth
    from nmrgnn_amd.graph import GraphBatch
    for pad in (0.05, 0.0):
        b = synth.make_batch(12, 256, 16, 10, pad, seed=5)
        gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
        y = torch.from_numpy(b["y"]).to(gpu_device); w = torch.from_numpy(b["w"]).to(gpu_device)
        ea, eb = _engines(64, gpu_device)
        eb.edge_table_force_fallback = True
        pa, e_a, ga = _step(ea, gb, y, w)
        pb, e_b, gb_ = _step(eb, gb, y, w)
        assert eb.last_table is not None and eb.edge_table_report(eb.last_table)[0]
        assert torch.equal(e_a, e_b) and torch.equal(pa, pb) and torch.equal(ga, gb_)
        assert torch.equal(ea.forward(gb), eb.forward(gb))


@pytest.mark.parametrize("scale,bias", [(1.0, 0.0), (4.0, 0.0), (4.0, 5.0), (16.0, -5.0), (16.0, 5.0)])
def test_sharp_edge_functions_meet_the_tolerance_or_raise_the_guard(gpu_device, scale, bias):
    """edge weights x 4 / x 16 and biases +- 5: the edge function gets steeper than anything glorot initialisation gives.
    Either the midpoint check passes and the table path agrees with the per-edge path to the tolerances of the ordinary case,
    or the guard is up and the results are the per-edge ones."""
    from nmrgnn_amd import synth
    from nmrgnn_amd.graph import GraphBatch
    b = synth.make_batch(12, 256, 16, 10, 0.05, seed=7)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    y = torch.from_numpy(b["y"]).to(gpu_device); w = torch.from_numpy(b["w"]).to(gpu_device)
    ea, eb = _engines(64, gpu_device)
    rng = np.random.default_rng(3)
    for eng in (ea, eb):
        sd = eng.params.state_dict()
        for t in range(4):
            sd[f"edge_fc/{t}/kernel"] = sd[f"edge_fc/{t}/kernel"] * np.float32(scale)
            sd[f"edge_fc/{t}/bias"] = (bias * np.random.default_rng(10 + t).uniform(-1, 1, sd[f"edge_fc/{t}/bias"].shape)).astype(np.float32)
        eng.params.load_state_dict(sd)
    pa, e_a, ga = _step(ea, gb, y, w)
    pb, e_b, gb_ = _step(eb, gb, y, w)
    up, err, sc = eb.edge_table_report(eb.last_table)
    print(f"[edge weights x {scale}, biases +-{bias}] guard {'UP' if up else 'down'}: midpoint error {err:.3e}, max |e| {sc:.3e}, "
          f"ratio {err / max(sc, 1e-30):.2e}")
    assert torch.isfinite(pb).all() and torch.isfinite(gb_).all()
    if up:
        assert torch.equal(e_a, e_b) and torch.equal(pa, pb) and torch.equal(ga, gb_)
    else:
        assert err <= eb.edge_table_tol * sc
        assert float((e_a - e_b).abs().max()) <= 4e-6 * max(float(e_a.abs().max()), 1.0)
        assert float((pa - pb).abs().max()) <= 4e-6 * max(float(pa.abs().max()), 1.0)
        for name in ea.params.offsets:
            x, z = ea.params.g(name), eb.params.g(name)
            assert float((x - z).abs().max()) <= 4e-5 * max(float(x.abs().max()), 1e-12), name


def test_table_is_kept_over_calls_while_the_weights_are_frozen(gpu_device):
    """inference with freeze_weights: the table is built by the first call (its range widened by a quarter) and later calls
    only check their distances against it; a frame whose distances leave the range is answered per edge (guard), a weight
    change rebuilds the table"""
    from nmrgnn_amd import synth
    from nmrgnn_amd.graph import GraphBatch
    ea, eb = _engines(64, gpu_device)
    ea.freeze_weights(True); eb.freeze_weights(True)
    batches = []
    for s, stretch in ((1, 1.0), (2, 1.02), (3, 0.97), (4, 3.0)):
        b = synth.make_batch(10, 256, 16, 10, 0.05, seed=20 + s)
        b["edges"] = (b["edges"] * np.float32(stretch)).astype(np.float32)
        batches.append(GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device))
    built = []
    for i, gb in enumerate(batches):
        pa, pb = ea.forward(gb), eb.forward(gb)
        built.append(id(eb._table_cache["e_all"]))
        assert float((pa - pb).abs().max()) <= 2e-6 * max(float(pa.abs().max()), 1.0), i
    assert len(set(built)) == 1                      # one table for the four calls
    # the stretched frame was answered per edge: bit for bit
    tb = eb._edge_table_build(batches[3], batches[3].live_edges(force=True), batches[3].live_edges(force=True)[2], False, False)
    assert bool(tb["gate"].cpu()[0] != 0)
    assert torch.equal(ea.forward(batches[3]), eb.forward(batches[3]))
    tb = eb._edge_table_build(batches[1], batches[1].live_edges(force=True), batches[1].live_edges(force=True)[2], False, False)
    assert bool(tb["gate"].cpu()[0] == 0)
    # new weights: a new table
    sd = eb.params.state_dict()
    sd["edge_fc/0/kernel"] = sd["edge_fc/0/kernel"] * np.float32(1.1)
    ea.params.load_state_dict(sd); eb.params.load_state_dict(sd)
    pa, pb = ea.forward(batches[0]), eb.forward(batches[0])
    assert id(eb._table_cache["e_all"]) != built[0]
    assert float((pa - pb).abs().max()) <= 2e-6 * max(float(pa.abs().max()), 1.0)


@pytest.mark.parametrize("cfg", [dict(atom_feature_size=32, edge_feature_size=2, edge_hidden_size=64, mp_layers=2, fc_layers=3, edge_fc_layers=3),
                                 dict(atom_feature_size=64, edge_feature_size=8, edge_hidden_size=32, mp_layers=2, fc_layers=2, edge_fc_layers=5),
                                 dict(atom_feature_size=64, edge_feature_size=3, edge_hidden_size=256, mp_layers=1, fc_layers=2, edge_fc_layers=2)])
def test_host_guarded_table_for_edge_shapes_without_the_fused_kernels(gpu_device, cfg):
    """edge_hidden_size != 128 / other depths: the layered edge MLP has no device-gated form, so the table's guard is read on the host
    (Engine.edge_table_sync): outputs and gradients against the per-edge engine; a forced guard gives the per-edge bits"""
    from nmrgnn_amd import synth
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
    from helpers import randomize_biases
    hp = declare_gnn_space(HyperParameters(**cfg))
    b = synth.make_batch(6, 200, 16, 10, 0.1, seed=21)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    y = torch.from_numpy(b["y"]).to(gpu_device); w = torch.from_numpy(b["w"]).to(gpu_device)
    ea, eb, ec = (Engine(hp, 10, device=gpu_device, seed=8) for _ in range(3))
    for e in (ea, eb, ec):
        randomize_biases(e)
        e.edge_table_min_edges = 0
    ea.edge_table = False
    ec.edge_table_force_fallback = True
    assert not ea.lib.ng_edge_live_supported(ea.H, ea.E, ea.Le, ea.fc_act)
    grads = []
    for e in (ea, eb, ec):
        p_inf = e.forward(gb).clone()
        p = e.forward(gb, training=True, seed=5)
        used = e.tape.table_sync is not None
        ee = e.tape.e.clone()
        l, d = e.loss_l2(gb, y, w, p)
        e.backward(d)
        grads.append((p_inf, p.clone(), ee, e.params.grad.clone(), used))
    torch.cuda.synchronize()
    (ia, pa, ea_e, ga, ua), (ib, pb, eb_e, gb_, ub), (ic, pc, ec_e, gc, uc) = grads
    assert not ua and ub and not uc
    assert torch.equal(ia, ic) and torch.equal(pa, pc) and torch.equal(ga, gc)          # guard forced up: the per-edge call
    scale = max(float(pa.abs().max()), 1.0)
    assert float((pa - pb).abs().max()) <= 4e-6 * scale and float((ia - ib).abs().max()) <= 4e-6 * scale
    assert float((ea_e - eb_e).abs().max()) <= 2e-6 * max(float(ea_e.abs().max()), 1.0)
    for name in ea.params.offsets:
        x, z = ea.params.g(name), eb.params.g(name)
        assert float((x - z).abs().max()) <= 4e-5 * max(float(x.abs().max()), 1e-12), name
