"""shared helpers for the parity tests (test infrastructure)"""
import numpy as np

from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
from nmrgnn_amd import synth


def make_hp(**kw):
    hp = HyperParameters(**kw)
    declare_gnn_space(hp)
    return hp


def hp_to_oracle(hp):
    from oracle import nmrgnn_oracle as O
    return O.hypers(**hp.as_dict())


def small_batch(n_graphs=3, n_atoms=50, K=16, num_elem=10, seed=7, p_pad=0.1):
    return synth.make_batch(n_graphs, n_atoms, K, num_elem, p_pad, seed)


def randomize_biases(engine, seed=3, scale=0.1):
    """non-zero biases so the bias / mask paths are exercised"""
    rng = np.random.default_rng(seed)
    sd = engine.params.state_dict()
    for k in sd:
        if k.endswith("bias"):
            sd[k] = (scale * rng.standard_normal(sd[k].shape)).astype(np.float32)
    engine.params.load_state_dict(sd)
    return sd


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
