"""Every selectable kernel variant must agree with the default path (and hence with the oracle):
NG_MP_PATH (split | fused | layered), NG_EDGE_BWD (v1 | v2), NG_EDGE_FWD (default | tm32),
NG_EDGE_PATH=layered, NG_DENSE_PATH=generic, NG_AGG_PATH=window."""
import numpy as np
import pytest

from helpers import make_hp, small_batch, randomize_biases, rel_err

pytestmark = pytest.mark.gpu

VARIANTS = [
    {"NG_MP_PATH": "fused"}, {"NG_MP_PATH": "layered"}, {"NG_EDGE_BWD": "v2"}, {"NG_EDGE_FWD": "tm32"},
    {"NG_EDGE_PATH": "layered"}, {"NG_DENSE_PATH": "generic"},
    {"NG_MP_PATH": "layered", "NG_AGG_PATH": "window"},
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
