"""GPU parity: the HIP path (through the C ABI) against the float64 oracle on the same inputs.
Tolerance: 1e-4 absolute on predicted shifts (BASELINE.json north_star), gradients 2e-4 relative
to the largest entry of each tensor."""
import numpy as np
import pytest

from helpers import make_hp, hp_to_oracle, small_batch, randomize_biases, rel_err

pytestmark = pytest.mark.gpu

PEAK_ATOL = 1e-4
GRAD_RTOL = 2e-4

CONFIGS = [
    dict(atom_feature_size=64, edge_feature_size=3, edge_hidden_size=128),   # bench arch
    dict(atom_feature_size=32, edge_feature_size=2, edge_hidden_size=32, mp_layers=2,
         fc_layers=3, edge_fc_layers=3),
    dict(atom_feature_size=256, edge_feature_size=3, edge_hidden_size=128),  # bundled-model arch
    dict(atom_feature_size=128, edge_feature_size=8, edge_hidden_size=64, mp_layers=1,
         fc_layers=2, edge_fc_layers=2),
]


def _setup(gpu_device, cfg, n_graphs=3, n_atoms=50, seed=7):
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    hp = make_hp(**cfg)
    b = small_batch(n_graphs, n_atoms, seed=seed)
    rng = np.random.default_rng(5)
    std = rng.uniform(0.5, 2.0, 10).astype(np.float32)
    avg = rng.uniform(-1.0, 1.0, 10).astype(np.float32)
    eng = Engine(hp, 10, std, avg, device=gpu_device, seed=11)
    sd = randomize_biases(eng)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"],
                    device=gpu_device)
    return hp, b, eng, sd, gb, std, avg


@pytest.mark.parametrize("cfg", CONFIGS)
def test_forward_inference_matches_oracle(gpu_device, cfg):
    import torch
    from oracle import nmrgnn_oracle as O
    hp, b, eng, sd, gb, std, avg = _setup(gpu_device, cfg)
    peaks = eng.forward(gb, training=False).cpu().numpy()
    ref = O.gnn_forward((b["atoms"], b["nlist"], b["edges"], b["inv_degree"]), sd, hp_to_oracle(hp),
                        std, avg)
    assert peaks.shape == ref.shape
    assert np.max(np.abs(peaks - ref)) < PEAK_ATOL, np.max(np.abs(peaks - ref))


@pytest.mark.parametrize("cfg", CONFIGS)
def test_training_forward_backward_matches_oracle(gpu_device, cfg):
    import torch
    from oracle import nmrgnn_oracle as O
    hp, b, eng, sd, gb, std, avg = _setup(gpu_device, cfg)
    N, K = b["edges"].shape
    Fh = hp.get('atom_feature_size') // 2
    xi = eng.randn(N * K, seed=123)
    mask = eng.dropout_mask(N * Fh, seed=321)
    peaks = eng.forward(gb, training=True, noise=xi, dropout_mask=mask)
    rng = np.random.default_rng(9)
    dpeaks = rng.standard_normal(N).astype(np.float32)
    eng.backward(torch.from_numpy(dpeaks).to(gpu_device))
    grads = eng.params.grads_dict()
    xi_h = xi.cpu().numpy().reshape(N, K)
    keep = 0.8
    mask_h = (mask.cpu().numpy().reshape(N, Fh) > 0).astype(np.float64)
    # sanity of the RNG kernels
    assert abs(xi_h.mean()) < 0.1 and abs(xi_h.std() - 1.0) < 0.1
    assert abs(mask_h.mean() - keep) < 0.05
    ref_peaks, ref_grads = O.gnn_forward_backward(
        (b["atoms"], b["nlist"], b["edges"], b["inv_degree"]), sd, hp_to_oracle(hp), dpeaks, std, avg,
        training=True, noise=xi_h, dropout_mask=mask_h)
    assert np.max(np.abs(peaks.cpu().numpy() - ref_peaks)) < PEAK_ATOL
    bad = {}
    for k, g in ref_grads.items():
        err = rel_err(grads[k], g)
        if err > GRAD_RTOL:
            bad[k] = err
    assert not bad, bad


def test_batch_invariance(gpu_device):
    """KAT-6: model(concat(g1,g2)) == concat(model(g1), model(g2))."""
    from nmrgnn_amd.graph import GraphBatch
    hp, b, eng, sd, gb, std, avg = _setup(gpu_device, CONFIGS[0], n_graphs=2, n_atoms=40)
    full = eng.forward(gb).cpu().numpy()
    n = 40
    for g in range(2):
        sl = slice(g * n, (g + 1) * n)
        sub = GraphBatch(b["atoms"][sl], b["nlist"][sl] - g * n, b["edges"][sl], b["inv_degree"][sl],
                         device=gpu_device)
        part = eng.forward(sub).cpu().numpy()
        np.testing.assert_allclose(part, full[sl], rtol=0, atol=2e-6)


def test_padded_slot_index_is_irrelevant(gpu_device):
    """KAT-3: changing nlist in a padded (edges == 0) slot must not change any peak."""
    from nmrgnn_amd.graph import GraphBatch
    hp, b, eng, sd, gb, std, avg = _setup(gpu_device, CONFIGS[0], n_graphs=1, n_atoms=64)
    base = eng.forward(gb).cpu().numpy()
    nl = b["nlist"].copy()
    pad = b["edges"] == 0
    assert pad.any()
    nl[pad] = 17
    gb2 = GraphBatch(b["atoms"], nl, b["edges"], b["inv_degree"], device=gpu_device)
    np.testing.assert_array_equal(eng.forward(gb2).cpu().numpy(), base)


def test_loss_and_adam_match_oracle(gpu_device):
    import torch
    from oracle import nmrgnn_oracle as O
    hp, b, eng, sd, gb, std, avg = _setup(gpu_device, CONFIGS[1])
    N = gb.N
    rng = np.random.default_rng(2)
    pred = rng.standard_normal(N).astype(np.float32)
    w = (rng.random(N) > 0.3).astype(np.float32)
    w[: gb.graph_ptr_host[1]] = 0.0          # a graph with zero total weight -> divide_no_nan
    y = b["y"]
    dev = gpu_device
    loss, dpred = eng.loss_l2(gb, torch.from_numpy(y).to(dev), torch.from_numpy(w).to(dev),
                              torch.from_numpy(pred).to(dev))
    rl, rg = O.batch_loss_s1(y, w, pred, gb.graph_ptr_host)
    assert abs(float(loss.cpu()) - rl) < 1e-5 * max(1.0, abs(rl))
    np.testing.assert_allclose(dpred.cpu().numpy(), rg, rtol=1e-5, atol=1e-7)
    # Adam: three steps on a fixed gradient
    p0 = eng.params.flat.cpu().numpy().astype(np.float64)
    g = rng.standard_normal(p0.shape).astype(np.float32)
    eng.params.grad.copy_(torch.from_numpy(g).to(dev))
    m = np.zeros_like(p0); v = np.zeros_like(p0); p = p0.copy()
    for t in range(1, 4):
        eng.adam_step(lr=1e-3)
        p, m, v = O.adam_step(p, g.astype(np.float64), m, v, t, lr=1e-3)
    np.testing.assert_allclose(eng.params.flat.cpu().numpy(), p, rtol=2e-5, atol=1e-7)
