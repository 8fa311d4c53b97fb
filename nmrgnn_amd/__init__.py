"""nmrgnn_amd — MI355X-native engine for nmrgnn's message-passing hot path.

Drop-in surface of the reference package (nmrgnn/__init__.py:21-31):
``load_model``, ``universe2graph``, ``check_peaks``, ``MPLayer``, ``RBFExpansion``, ``EdgeFCBlock``,
``MPBlock``, ``FCBlock``, ``GNNModel``, ``build_GNNModel``, ``NameLoss``.
The compute path is libnmrgnn_hip.so (hand-written gfx950 HIP kernels) reached through ctypes;
there is no CPU fallback.
"""
__version__ = "0.1.0"

from .hypers import HyperParameters, declare_gnn_space  # noqa: F401

_LAZY = {
    "Engine": "engine", "GraphBatch": "graph", "concat_graphs": "graph", "BatchPrefetcher": "graph",
    "MPLayer": "layers", "AMPLayer": "layers", "RBFExpansion": "layers", "EdgeFCBlock": "layers", "MPBlock": "layers",
    "FCBlock": "layers", "GNNModel": "model", "build_GNNModel": "model",
    "load_model": "library", "universe2graph": "library", "check_peaks": "library",
    "save_model": "library", "NameLoss": "losses", "Trainer": "train",
}


def __getattr__(name):
    mod = _LAZY.get(name)
    if mod is None:
        raise AttributeError(f"module 'nmrgnn_amd' has no attribute {name!r}")
    import importlib
    return getattr(importlib.import_module(f".{mod}", __name__), name)
