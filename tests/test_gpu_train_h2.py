"""Training with the split-operand edge kernels follows the f32-input-MFMA training step for step: same seeds
(noise, dropout, weights), 12 Adam steps on a 64-graph batch; the per-step losses and the final weights must coincide
to fp32 rounding accumulated over the steps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(gpu_device, monkeypatch, math, steps=12):
    import torch
    from nmrgnn_amd import synth
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
    from nmrgnn_amd.train import Trainer
    monkeypatch.setenv("NG_EDGE_MATH", math)
    hp = declare_gnn_space(HyperParameters(atom_feature_size=64, edge_feature_size=3, edge_hidden_size=128, mp_layers=4,
                                           fc_layers=4, edge_fc_layers=4))
    eng = Engine(hp, 10, device=gpu_device, seed=1234)
    b = synth.make_batch(64, 256, 16, 10, 0.05, seed=42)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    gb.csc()
    y = torch.from_numpy(b["y"]).to(gpu_device)
    w = torch.from_numpy(b["w"]).to(gpu_device)
    tr = Trainer(eng, lr=1e-3)
    losses = [float(tr.step(gb, y, w)) for _ in range(steps)]
    return np.array(losses), eng.params.flat.detach().cpu().numpy().astype(np.float64)


def test_training_trajectories_coincide(gpu_device, monkeypatch):
    """f16x2: the default two-piece fp16 kernels (edge_fwd_h2 / edge_bwd_h2)"""
    l_h2, p_h2 = _run(gpu_device, monkeypatch, "f16x2")
    l_32, p_32 = _run(gpu_device, monkeypatch, "fp32")
    assert l_32[-1] < l_32[0]                                   # it trains
    assert np.max(np.abs(l_h2 - l_32) / np.abs(l_32)) < 2e-5, (l_h2, l_32)
    # Adam normalises every update to ~lr, so rounding-level gradient differences move a weight by << lr per step
    assert np.max(np.abs(p_h2 - p_32)) < 12 * 1e-3 * 0.05
