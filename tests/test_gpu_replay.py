"""Graph replay of small calls (nmrgnn_amd/replay.py, ABI 7: ng_replay_arm / ng_replay_stage): the replayed chain gives the
bits of the eager chain — parameter trajectories of one-graph training steps, peaks of one-frame forwards."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graphs(n, atoms=256, K=16):
    from nmrgnn_amd import synth
    out = []
    for g in range(n):
        b = synth.make_batch(1, atoms, K, 10, 0.05, seed=500 + g)
        out.append(((b["atoms"], b["nlist"], b["edges"], b["inv_degree"]), b["graph_ptr"], b["y"], b["w"]))
    return out


@pytest.mark.parametrize("F", [64, 256])
def test_train_step_replay_is_the_eager_trajectory(gpu_device, F):
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
    from nmrgnn_amd.replay import TrainStepReplay
    from nmrgnn_amd.train import Trainer
    hp = declare_gnn_space(HyperParameters(atom_feature_size=F, edge_feature_size=3, edge_hidden_size=128, mp_layers=4,
                                           fc_layers=4, edge_fc_layers=4))
    gs = _graphs(6)
    dev = gpu_device
    t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a)).to(device=dev, dtype=dt)
    ea, eb = Engine(hp, 10, device=dev, seed=77), Engine(hp, 10, device=dev, seed=77)
    assert torch.equal(ea.params.flat, eb.params.flat)
    ta, tb = Trainer(ea, lr=1e-3), Trainer(eb, lr=1e-3)
    raw0, gp0, y0, w0 = gs[0]
    rp = TrainStepReplay(tb, raw0, y0, w0, graph_ptr=gp0)
    assert torch.equal(ea.params.flat, eb.params.flat) and eb.adam_t == 0     # the capture left the training state alone
    assert torch.equal(ea.adam_m, eb.adam_m) and torch.equal(ea.adam_v, eb.adam_v)
    for step in range(7):
        raw, gp, y, w = gs[step % len(gs)]
        la = ta.step(GraphBatch(*raw, graph_ptr=gp, device=dev), t(y), t(w))
        lb = rp.step(raw, y, w)
        torch.cuda.synchronize()
        assert torch.equal(la, lb), (step, la, lb)
        assert torch.equal(ea.params.flat, eb.params.flat), step
    assert torch.equal(ea.adam_m, eb.adam_m) and torch.equal(ea.adam_v, eb.adam_v)
    assert ea.adam_t == eb.adam_t == 7
    # and an eager step on the replayed engine continues the same trajectory (packed images, step counters, seeds)
    raw, gp, y, w = gs[1]
    la = ta.step(GraphBatch(*raw, graph_ptr=gp, device=dev), t(y), t(w))
    lb = tb.step(GraphBatch(*raw, graph_ptr=gp, device=dev), t(y), t(w))
    assert torch.equal(la, lb) and torch.equal(ea.params.flat, eb.params.flat)


def test_forward_replay_is_the_eager_forward(gpu_device):
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import frames_to_batch
    from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
    from nmrgnn_amd.replay import ForwardReplay
    from nmrgnn_amd.structure import atoms_onehot, read_pdb
    s = read_pdb(os.path.join(os.path.dirname(__file__), "data", "108M.pdb"))
    atoms = atoms_onehot(s.elements)
    rng = np.random.default_rng(3)
    frames = [s.frames[0] + rng.normal(0, 0.2, s.frames[0].shape).astype(np.float32) for _ in range(4)]
    eng = Engine(declare_gnn_space(HyperParameters()), atoms.shape[1], device=gpu_device, seed=5)
    eng.freeze_weights(True)
    rp = ForwardReplay(eng, atoms, frames[0])
    at = torch.from_numpy(atoms).to(gpu_device)
    for f in frames[::-1]:
        ref = eng.forward(frames_to_batch(at, torch.from_numpy(f).to(gpu_device)[None], 16, device=gpu_device)).clone()
        got = rp(f)
        torch.cuda.synchronize()
        assert torch.equal(ref, got)
