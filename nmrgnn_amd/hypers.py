"""Hyper-parameter container mirroring the subset of ``kerastuner.HyperParameters`` that the
reference touches (nmrgnn/model.py:22-36: Choice / Int / Fixed / get).  keras-tuner itself (the HPO
harness) is out of scope; only names, defaults and the ``hypers.get(name)`` accessor matter to the
hot path because they decide every tensor shape."""
from __future__ import annotations


class HyperParameters:
    def __init__(self, **values):
        self.values = {}
        self.space = {}
        self.values.update(values)

    # -- the three declaration forms used by build_GNNModel (model.py:22-36,45)
    def Choice(self, name, values, ordered=None, default=None):
        self.space[name] = ("choice", list(values))
        if name not in self.values:
            self.values[name] = values[0] if default is None else default
        elif self.values[name] not in values:
            raise ValueError(f"{name}={self.values[name]!r} is not one of {values}")
        return self.values[name]

    def Int(self, name, min_value, max_value, step=1, default=None):
        self.space[name] = ("int", (min_value, max_value, step))
        if name not in self.values:
            self.values[name] = min_value if default is None else default
        v = self.values[name]
        if not (min_value <= v <= max_value):
            raise ValueError(f"{name}={v} outside [{min_value},{max_value}]")
        return v

    def Fixed(self, name, value):
        self.space[name] = ("fixed", value)
        self.values.setdefault(name, value)
        return self.values[name]

    def get(self, name):
        if name not in self.values:
            raise KeyError(f"{name} does not exist")
        return self.values[name]

    def __contains__(self, name):
        return name in self.values

    def as_dict(self):
        return dict(self.values)

    def __repr__(self):
        return f"HyperParameters({self.values})"


def declare_gnn_space(hp: HyperParameters) -> HyperParameters:
    """The search space / defaults of nmrgnn/model.py:22-36 and the learning rate of model.py:45."""
    hp.Choice('atom_feature_size', [32, 64, 128, 256], ordered=True, default=256)
    hp.Choice('edge_feature_size', [1, 2, 3, 8, 64], ordered=True, default=3)
    hp.Choice('edge_hidden_size', [16, 32, 64, 128, 256], ordered=True, default=128)
    hp.Int('mp_layers', 1, 6, step=1, default=4)
    hp.Int('fc_layers', 2, 6, step=1, default=4)
    hp.Int('edge_fc_layers', 2, 6, step=1, default=4)
    hp.Choice('noise', [0.0, 0.025, 0.05, 0.1], ordered=True, default=0.025)
    hp.Choice('dropout', [True, False], default=True)
    hp.Fixed('rbf_low', 0.005)
    hp.Fixed('rbf_high', 0.20)
    hp.Choice('mp_activation', ['relu', 'softplus', 'tanh'], default='softplus')
    hp.Choice('fc_activation', ['relu', 'softplus'], default='softplus')
    hp.Choice('learning_rate', [1e-3, 5e-4, 1e-4, 1e-5], default=1e-4)
    return hp
