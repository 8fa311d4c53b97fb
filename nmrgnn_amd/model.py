"""``GNNModel`` / ``build_GNNModel`` — the reference's model object (nmrgnn/model.py:12-105, 205-274)
on top of the HIP engine.  ``model((atoms, nlist, edges, inv_degree), training=False)`` returns one
chemical shift per atom, exactly like the Keras model's ``call``.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .engine import Engine
from .graph import GraphBatch
from .hypers import HyperParameters, declare_gnn_space
from .standards import load_standards

LOTS_OF_ELEMENTS = 100   # nmrgnn/model.py:222


class Peaks(np.ndarray):
    """numpy array that also answers ``.numpy()`` like the TF eager tensor the reference returns."""

    def numpy(self):
        return np.asarray(self)


class GNNModel:
    def __init__(self, hypers, peak_standards, name='gnn-model', device=None, seed=1234, **kwargs):
        self.hypers = hypers
        self.name = name
        self.peak_standards = dict(peak_standards)
        # nmrgnn/model.py:220-228: large std/avg tables, cut to num_elem at build time
        self.peak_std = np.ones(LOTS_OF_ELEMENTS, dtype=np.float32)
        self.peak_avg = np.zeros(LOTS_OF_ELEMENTS, dtype=np.float32)
        for k, v in self.peak_standards.items():
            self.peak_std[k] = v[2]
            self.peak_avg[k] = v[1]
        self.embed_dim = hypers.get('atom_feature_size')
        self._device = device
        self._seed = seed
        self.engine = None
        self._pending_state = None
        self._train_calls = 0         # training-mode calls so far: every one draws fresh noise / dropout
        self._flat_leaf = None        # torch.nn.Parameter over the flat weight buffer, created by parameters()

    # -- keras-like build: the number of elements comes from the first input (model.py:236-243)
    def build(self, num_elem):
        if self.engine is not None:
            if self.engine.C != num_elem:
                raise ValueError(f"model was built for {self.engine.C} elements, got {num_elem}")
            return
        self.engine = Engine(self.hypers, num_elem, self.peak_std[:num_elem], self.peak_avg[:num_elem],
                             device=self._device, seed=self._seed)
        if self._pending_state is not None:
            self.engine.params.load_state_dict(self._pending_state)
            self._pending_state = None
        self._restore_adam()

    def _restore_adam(self):
        pend = getattr(self, "_pending_adam", None)
        if pend is None or self.engine is None:
            return
        self._pending_adam = None
        if pend == "reset":            # a checkpoint without optimiser state: moments of OTHER weights must not survive
            self.engine.adam_m.zero_()
            self.engine.adam_v.zero_()
            self.engine.adam_t = 0
            return
        m, v, t = pend
        if m.shape[0] != self.engine.adam_m.shape[0]:
            raise ValueError(f"checkpoint Adam state has {m.shape[0]} entries, the model's flat parameter buffer "
                             f"{self.engine.adam_m.shape[0]}: it belongs to a different architecture")
        self.engine.adam_m.copy_(torch.from_numpy(m))
        self.engine.adam_v.copy_(torch.from_numpy(v))
        self.engine.adam_t = t

    def _as_batch(self, inputs):
        if isinstance(inputs, GraphBatch):
            return inputs, True
        atoms, nlist, edges, inv_degree = inputs
        on_device = all(isinstance(x, torch.Tensor) and x.is_cuda for x in (atoms, nlist, edges))
        num_elem = int(atoms.shape[-1])
        self.build(num_elem)
        return GraphBatch(atoms, nlist, edges, inv_degree, device=self.engine.device), on_device

    def __call__(self, inputs, training=False, seed=None):
        """``training=True`` applies GaussianNoise and Dropout with a FRESH draw per call, as Keras does
        (model.py:253,266-267): the Philox key is derived from the model seed and a per-model call counter;
        pass ``seed=`` to fix it."""
        batch, on_device = self._as_batch(inputs)
        if self.engine is None:
            self.build(batch.C)
        if training and seed is None:
            seed = ((int(self._seed) * 0x9E3779B97F4A7C15) ^ (self._train_calls * 1000003 + 0x632BE5AB)) & ((1 << 63) - 1)
            self._train_calls += 1
        leaf = self._flat_leaf
        if leaf is not None and leaf.requires_grad and torch.is_grad_enabled():
            # differentiable call (nmrgnn/main.py:74-80 trains by autodiff through model(x)): the result is a device
            # tensor with a grad_fn whatever the input container was — a numpy array could not carry one
            from .autograd import model_forward
            return model_forward(self.engine, leaf, batch, training=training, seed=seed or 0)
        peaks = self.engine.forward(batch, training=training, seed=seed or 0)
        if on_device:
            return peaks
        return peaks.cpu().numpy().view(Peaks)

    call = __call__
    predict = __call__

    # -- torch.autograd surface (the outer autograd shell)
    def parameters(self):
        """The trainable state as ONE leaf tensor: the engine's flat fp32 parameter buffer, shared, not copied.
        After this call ``model(g)`` under ``torch.enable_grad()`` returns a tensor with a grad_fn, and
        ``loss.backward()`` accumulates into ``parameters()[0].grad`` (autograd.GNNModelFunction).
        ``torch.optim.Adam(model.parameters(), lr, eps=1e-7)`` is the reference's optimiser (model.py:44-45)."""
        self._need_engine()
        if self._flat_leaf is None:
            self._flat_leaf = torch.nn.Parameter(self.engine.params.flat, requires_grad=True)
        return [self._flat_leaf]

    def named_parameter_views(self):
        """{keras-style name: (weight view, gradient view or None)} into the flat leaf and its ``.grad``"""
        leaf = self.parameters()[0]
        P = self.engine.params
        out = {}
        for name, shape in P.shapes.items():
            o, n = P.offsets[name], int(np.prod(shape))
            g = None if leaf.grad is None else leaf.grad[o:o + n].view(*shape)
            out[name] = (leaf.data[o:o + n].view(*shape), g)
        return out

    def requires_grad_(self, on=True):
        self.parameters()[0].requires_grad_(bool(on))
        return self

    def freeze(self, on=True):
        """declare the weights constant (inference): packed weight images are cached across calls"""
        self._need_engine()
        self.engine.freeze_weights(on)
        return self

    # -- weights
    def get_weights(self):
        self._need_engine()
        return self.engine.params.state_dict()

    def set_weights(self, state):
        if self.engine is None:
            self._pending_state = state
        else:
            self.engine.params.load_state_dict(state)

    def count_params(self):
        self._need_engine()
        return self.engine.params.count()

    def _need_engine(self):
        if self.engine is None:
            raise RuntimeError("model is not built yet: call it once (or model.build(num_elem))")

    def get_config(self):
        return {'hypers': self.hypers.as_dict(), 'peak_standards': self.peak_standards}

    def save(self, path):
        """own flat format: <path>/weights.npz + <path>/config.json (names follow the Keras variable tree).
        weights.npz also carries the Adam state (``__adam_m`` / ``__adam_v`` flat buffers, ``__adam_t``) once a
        step has been taken, so training resumes where it stopped (the reference's checkpoints hold the Adam
        slots too, main.py:63-68).  The TensorFlow bundle written beside it holds the WEIGHT tensors under the
        reference's variable keys — it is what ``load_model`` reads back and what a TF-side reader can pick tensors
        from by key; it is not a complete Keras SavedModel (no object graph, no saved_model.pb)."""
        self._need_engine()
        os.makedirs(path, exist_ok=True)
        arrays = {k.replace("/", "."): v for k, v in self.engine.params.state_dict().items()}
        if self.engine.adam_t > 0:
            arrays["__adam_m"] = self.engine.adam_m.cpu().numpy()
            arrays["__adam_v"] = self.engine.adam_v.cpu().numpy()
            arrays["__adam_t"] = np.asarray(self.engine.adam_t, np.int64)
        arrays["__train_calls"] = np.asarray(self._train_calls, np.int64)   # seeds the next noise / dropout draw
        np.savez(os.path.join(path, "weights.npz"), **arrays)
        cfg = {"hypers": self.hypers.as_dict(), "num_elem": self.engine.C,
               "peak_standards": {str(k): list(v) for k, v in self.peak_standards.items()}}
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(cfg, f, indent=1)
        # the same weights as a TensorFlow checkpoint bundle under the reference's variable names
        from .tfbundle import save_gnn_bundle
        save_gnn_bundle(os.path.join(path, "variables", "variables"), self.engine.params.state_dict(),
                        self.hypers)

    def load_weights(self, path):
        """weights.npz (own format) or a TensorFlow checkpoint bundle / SavedModel directory."""
        from .library import _bundle_prefix
        if not path.endswith(".npz") and not os.path.exists(os.path.join(path, "weights.npz")):
            prefix = _bundle_prefix(path)
            if prefix is None:
                raise ValueError(f"{path}: no weights.npz and no checkpoint bundle")
            from .tfbundle import load_gnn_bundle
            self.set_weights(load_gnn_bundle(prefix)[0])
            self._pending_adam = "reset"       # our bundle reader takes the weight tensors only
            self._restore_adam()
            return
        f = path if path.endswith(".npz") else os.path.join(path, "weights.npz")
        z = np.load(f)
        self.set_weights({k.replace(".", "/"): z[k] for k in z.files if not k.startswith("__")})
        if "__adam_t" in z.files:
            self._pending_adam = (z["__adam_m"], z["__adam_v"], int(z["__adam_t"]))
        else:
            self._pending_adam = "reset"
        if "__train_calls" in z.files:
            self._train_calls = int(z["__train_calls"])
        self._restore_adam()


def build_GNNModel(hp=None, metrics=True, loss_balance=1.0, device=None):
    """nmrgnn/model.py:12-105.  Declares the hyper-parameter space on ``hp`` (same names/defaults),
    loads the peak standards and returns a GNNModel with ``optimizer`` / ``loss`` attributes set
    (Adam at hp['learning_rate'], NameLoss(s=loss_balance)).  The 15 keras metrics of the
    reference are training observability and are not part of the engine."""
    from .losses import NameLoss
    hp = HyperParameters() if hp is None else hp
    declare_gnn_space(hp)
    model = GNNModel(hp, load_standards(), device=device)
    model.loss = NameLoss(label_idx=None, s=loss_balance)
    model.optimizer = {"name": "Adam", "learning_rate": hp.get('learning_rate'),
                       "beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-7}
    return model
