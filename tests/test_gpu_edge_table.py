"""Opt-in edge path through a table of the edge function (csrc/edge_table.hip, Engine.edge_table): e, peaks and every
gradient against the per-edge kernels and the float64 oracle; run-to-run bits."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engines(F, dev, seed=3):
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
    hp = declare_gnn_space(HyperParameters(atom_feature_size=F, edge_feature_size=3, edge_hidden_size=128, mp_layers=4,
                                           fc_layers=4, edge_fc_layers=4))
    a, b = Engine(hp, 10, device=dev, seed=seed), Engine(hp, 10, device=dev, seed=seed)
    b.edge_table = True
    return a, b


@pytest.mark.parametrize("F,graphs", [(64, 24), (256, 6)])
def test_table_path_matches_the_per_edge_path(gpu_device, F, graphs):
    from nmrgnn_amd import synth
    from nmrgnn_amd.graph import GraphBatch
    b = synth.make_batch(graphs, 256, 16, 10, 0.05, seed=11)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    y = torch.from_numpy(b["y"]).to(gpu_device); w = torch.from_numpy(b["w"]).to(gpu_device)
    ea, eb = _engines(F, gpu_device)
    for training in (False, True):
        pa = ea.forward(gb, training=training, seed=99)
        pb = eb.forward(gb, training=training, seed=99)
        scale = float(pa.abs().max())
        assert float((pa - pb).abs().max()) <= 2e-6 * max(scale, 1.0), (training, float((pa - pb).abs().max()), scale)
    # edge features themselves: interpolation error far below fp32 resolution of the values
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


def test_table_path_against_the_oracle(gpu_device):
    """the tolerances the per-edge path is held to (tests/test_gpu_parity.py): 1e-4 on shifts, 2e-4 of the largest entry on every
    gradient tensor, against the float64 oracle — training mode with explicit noise and dropout draws"""
    from oracle import nmrgnn_oracle as O
    from helpers import hp_to_oracle, rel_err
    import test_gpu_parity as TP
    hp, b, eng, sd, gb, std, avg = TP._setup(gpu_device, TP.CONFIGS[0])
    eng.edge_table = True
    N, K = b["edges"].shape
    Fh = hp.get('atom_feature_size') // 2
    xi = eng.randn(N * K, seed=123)
    mask = eng.dropout_mask(N * Fh, seed=321)
    peaks = eng.forward(gb, training=True, noise=xi, dropout_mask=mask)
    assert eng.tape.table is not None
    dpeaks = np.random.default_rng(9).standard_normal(N).astype(np.float32)
    eng.backward(torch.from_numpy(dpeaks).to(gpu_device))
    grads = eng.params.grads_dict()
    ref_peaks, ref_grads = O.gnn_forward_backward(
        (b["atoms"], b["nlist"], b["edges"], b["inv_degree"]), sd, hp_to_oracle(hp), dpeaks, std, avg,
        training=True, noise=xi.cpu().numpy().reshape(N, K), dropout_mask=(mask.cpu().numpy().reshape(N, Fh) > 0).astype(np.float64))
    assert np.max(np.abs(peaks.cpu().numpy() - ref_peaks)) < TP.PEAK_ATOL
    bad = {k: rel_err(grads[k], g) for k, g in ref_grads.items() if rel_err(grads[k], g) > TP.GRAD_RTOL}
    assert not bad, bad
