"""Layer classes of the reference's public surface (nmrgnn/__init__.py:21-31), backed by the HIP
engine: ``MPLayer``, ``RBFExpansion``, ``EdgeFCBlock``, ``MPBlock``, ``FCBlock``.

They keep the reference constructors and call conventions (inputs may be numpy arrays, torch tensors
or anything exposing ``__array__``; weights are created on first call from the input shapes with the
Keras default initialisers: GlorotUniform kernels, zero biases).  They are forward-only building
blocks; training goes through :class:`nmrgnn_amd.model.GNNModel`, whose engine owns one flat
parameter buffer and the hand-derived backward.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import ptr
from .engine import ACT, rbf_grid


def _device():
    if not torch.cuda.is_available():
        raise _lib.NGError("nmrgnn_amd layers need an AMD GPU; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _f32(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32)), device=dev)


def _i32(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=torch.int32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x).astype(np.int32)), device=dev)


def _glorot(shape, dev, gen):
    if len(shape) == 2:
        fi, fo = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fi, fo = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fi + fo))
    return ((torch.rand(*shape, generator=gen) * 2 - 1) * lim).to(dev)


def get_regularizer(spec):
    """tf.keras.regularizers.get for the cases a weight regulariser can take without Keras: None, a callable
    (w -> scalar), 'l1' / 'l2' / 'l1_l2' (Keras default factor 0.01), or a Keras-style config dict
    {'class_name': 'L1L2', 'config': {'l1': .., 'l2': ..}}.  Anything else raises, as Keras does."""
    if spec is None or callable(spec):
        return spec
    l1 = l2 = 0.0
    if isinstance(spec, str):
        name = spec.lower()
        if name not in ("l1", "l2", "l1_l2"):
            raise ValueError(f"Could not interpret regularizer identifier: {spec!r}")
        l1 = 0.01 if name in ("l1", "l1_l2") else 0.0
        l2 = 0.01 if name in ("l2", "l1_l2") else 0.0
    elif isinstance(spec, dict):
        cfg = spec.get("config", spec)
        l1, l2 = float(cfg.get("l1", 0.0) or 0.0), float(cfg.get("l2", 0.0) or 0.0)
    else:
        raise ValueError(f"Could not interpret regularizer identifier: {spec!r}")

    def reg(w):
        out = 0.0
        if l1:
            out = out + l1 * w.abs().sum()
        if l2:
            out = out + l2 * (w * w).sum()
        return out
    return reg


class _Layer:
    _seed = 0

    def __init__(self):
        self.built = False
        _Layer._seed += 1
        self._gen = torch.Generator().manual_seed(1234 + _Layer._seed)

    def _ctx(self, dev):
        return _lib.get_context(dev.index)

    @staticmethod
    def _st(dev):
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class RBFExpansion(_Layer):
    """nmrgnn/layers.py:102-140: exp(-(d - mu_k)^2 / gap) on an evenly spaced grid of ``count`` centres."""

    def __init__(self, low, high, count, name='rbf-layer', **kwargs):
        super().__init__()
        self.low, self.high, self.count, self.name = low, high, count, name

    def get_config(self):
        return {'low': self.low, 'high': self.high, 'count': self.count, 'name': self.name}

    def __call__(self, inputs):
        dev = _device()
        d = _f32(inputs, dev)
        if self.count % 4:
            raise ValueError("RBFExpansion: count must be a multiple of 4")
        c, gap = rbf_grid(self.low, self.high, self.count)
        self.centers, self.gap = torch.from_numpy(c).to(dev), gap
        flat = d.reshape(-1)
        ones = torch.ones_like(flat)           # no mask here: the layer alone never zeroes
        out = torch.empty(flat.numel(), self.count, dtype=torch.float32, device=dev)
        ctx = self._ctx(dev)
        ctx.check(ctx.lib.ng_rbf_expand(ctx.handle, self._st(dev), flat.numel(), self.count, ptr(ones),
                                        ptr(flat), ptr(self.centers), self.gap, ptr(out)), "ng_rbf_expand")
        return out.reshape(*d.shape, self.count)


class MPLayer(_Layer):
    """nmrgnn/layers.py:5-46: out = activation(einsum('ijn,ijl,lmn,i->im', edges, nodes[nlist], w, inv_degree))."""

    def __init__(self, activation=None, kernel_regularizer=None, name='MPLayer', **kwargs):
        super().__init__()
        if activation not in ACT:
            raise ValueError(f"unsupported activation {activation!r}")
        self.activation, self.name = activation, name
        # layers.py:9,44-45: every call adds regularizer(w) to the layer's losses (keras add_loss)
        self.mpl_regularizer = get_regularizer(kernel_regularizer)
        self.losses = []
        self.w = None

    def get_config(self):
        return {'activation': self.activation, 'name': self.name}

    def build(self, F, E, dev):
        self.w = _glorot((F, F, E), dev, self._gen)   # keras add_weight default: glorot_uniform
        self.built = True

    def __call__(self, inputs, residual=False):
        nodes, nlist, edges, inv_degree = inputs
        dev = _device()
        nodes, nlist = _f32(nodes, dev), _i32(nlist, dev)
        edges, inv = _f32(edges, dev), _f32(inv_degree, dev).reshape(-1)
        N, F = nodes.shape
        K, E = nlist.shape[1], edges.shape[-1]
        if not self.built:
            self.build(F, E, dev)
        out = torch.empty(N, F, dtype=torch.float32, device=dev)
        ctx = self._ctx(dev)
        ctx.check(ctx.lib.ng_mp_layer_fwd(ctx.handle, self._st(dev), N, K, F, E, ACT[self.activation],
                                          1 if residual else 0, ptr(nodes), ptr(nlist), ptr(edges),
                                          ptr(inv), ptr(self.w), ptr(out), None, None),
                  "ng_mp_layer_fwd")
        if self.mpl_regularizer is not None:          # layers.py:44-45
            self.losses = [self.mpl_regularizer(self.w)]
        return out


class _DenseStack(_Layer):
    """sequence of keras Dense layers evaluated with ng_dense_fwd (shapes padded to the kernel's
    granularity: contraction % 8, outputs % 4 — zero padding does not change the product)."""

    def _dense(self, x, W, b, act, residual, dev):
        M, Kin = x.shape
        Nout = W.shape[1]
        Kp, Np = (Kin + 7) // 8 * 8, (Nout + 3) // 4 * 4
        if Kp != Kin:
            x = torch.nn.functional.pad(x, (0, Kp - Kin))
            W = torch.nn.functional.pad(W, (0, 0, 0, Kp - Kin))
        if Np != Nout:
            W = torch.nn.functional.pad(W, (0, Np - Nout))
            b = torch.nn.functional.pad(b, (0, Np - Nout))
        x, W, b = x.contiguous(), W.contiguous(), b.contiguous()
        y = torch.empty(M, Np, dtype=torch.float32, device=dev)
        ctx = self._ctx(dev)
        ctx.check(ctx.lib.ng_dense_fwd(ctx.handle, self._st(dev), M, Kp, Np, ACT[act],
                                       1 if (residual and Kp == Np) else 0, ptr(x), ptr(W), ptr(b),
                                       ptr(y), None), "ng_dense_fwd")
        return y[:, :Nout].contiguous() if Np != Nout else y

    def _dense_bwd(self, x, W, y, dy, act, dev):
        """gradients of ``y = act(x @ W + b)`` (no residual): (dx, dW, db); same zero padding as :meth:`_dense`."""
        M, Kin = x.shape
        Nout = W.shape[1]
        Kp, Np = (Kin + 7) // 8 * 8, (Nout + 3) // 4 * 4
        pad = torch.nn.functional.pad
        x, W = pad(x, (0, Kp - Kin)).contiguous(), pad(W, (0, Np - Nout, 0, Kp - Kin)).contiguous()
        y, dy = pad(y, (0, Np - Nout)).contiguous(), pad(dy, (0, Np - Nout)).contiguous()
        dx = torch.empty(M, Kp, dtype=torch.float32, device=dev)
        dW = torch.empty(Kp, Np, dtype=torch.float32, device=dev)
        db = torch.empty(Np, dtype=torch.float32, device=dev)
        ctx = self._ctx(dev)
        ctx.check(ctx.lib.ng_dense_bwd(ctx.handle, self._st(dev), M, Kp, Np, ACT[act], 0, ptr(x), ptr(W), ptr(y),
                                       ptr(dy), ptr(dx), ptr(dW), ptr(db)), "ng_dense_bwd")
        return dx[:, :Kin].contiguous(), dW[:Kin, :Nout].contiguous(), db[:Nout].contiguous()


class EdgeFCBlock(_DenseStack):
    """nmrgnn/model.py:109-144: (edge_fc_layers-1) x Dense(edge_hidden_size, fc_activation) + Dense(edge_feature_size)."""

    def __init__(self, hypers):
        super().__init__()
        self.hypers = hypers
        self.name = 'edge-fc-block'
        self.weights = None

    def __call__(self, edge_input):
        dev = _device()
        x = _f32(edge_input, dev)
        lead, D = x.shape[:-1], x.shape[-1]
        H, E = self.hypers.get('edge_hidden_size'), self.hypers.get('edge_feature_size')
        Le, act = self.hypers.get('edge_fc_layers'), self.hypers.get('fc_activation')
        if not self.built:
            dims = [D] + [H] * (Le - 1) + [E]
            self.weights = [(_glorot((dims[i], dims[i + 1]), dev, self._gen),
                             torch.zeros(dims[i + 1], device=dev)) for i in range(Le)]
            self.built = True
        x = x.reshape(-1, D)
        for i, (W, b) in enumerate(self.weights):
            x = self._dense(x, W, b, act if i < Le - 1 else None, False, dev)
        return x.reshape(*lead, E)

    def get_config(self):
        return {'hypers': self.hypers}


class MPBlock(_Layer):
    """nmrgnn/model.py:147-175: mp_layers x (nodes = MPLayer(nodes, ...) + nodes)."""

    def __init__(self, hypers):
        super().__init__()
        self.hypers = hypers
        self.name = 'mp-block'
        self.mp = [MPLayer(hypers.get('mp_activation')) for _ in range(hypers.get('mp_layers'))]

    def __call__(self, inputs):
        nodes = inputs[0]
        for layer in self.mp:
            nodes = layer([nodes] + list(inputs[1:]), residual=True)
        return nodes

    def get_config(self):
        return {'hypers': self.hypers}


class FCBlock(_DenseStack):
    """nmrgnn/model.py:178-202: (fc_layers-1) x residual Dense(F) then Dense(F//2), all fc_activation."""

    def __init__(self, hypers):
        super().__init__()
        self.hypers = hypers
        self.name = 'fc-block'
        self.weights = None

    def __call__(self, nodes):
        dev = _device()
        x = _f32(nodes, dev)
        F = self.hypers.get('atom_feature_size')
        Lf, act = self.hypers.get('fc_layers'), self.hypers.get('fc_activation')
        if x.shape[-1] != F:
            raise ValueError(f"FCBlock expects {F} features, got {x.shape[-1]}")
        if not self.built:
            dims = [F] * Lf + [F // 2]
            self.weights = [(_glorot((dims[i], dims[i + 1]), dev, self._gen),
                             torch.zeros(dims[i + 1], device=dev)) for i in range(Lf)]
            self.built = True
        for i, (W, b) in enumerate(self.weights):
            x = self._dense(x, W, b, act, i < Lf - 1, dev)
        return x

    def get_config(self):
        return {'hypers': self.hypers}


class AMPLayer(_DenseStack):
    """nmrgnn/layers.py:48-100: attention message passing.  ``b = softmax_j(inv_degree_i *
    <edges_ij @ wk, nodes_i @ wq>)``, ``out = activation(sum_j b_ij * (nodes[nlist_ij] @ wv))``.
    The reference only calls it in a shape test (the model is built from MPLayer); :meth:`backward` returns what
    TensorFlow's autodiff would for that call."""

    def __init__(self, activation=None, kernel_regularizer=None, name='AMPLayer', **kwargs):
        super().__init__()
        if activation not in ACT:
            raise ValueError(f"unsupported activation {activation!r}")
        self.activation, self.name = activation, name
        self.mpl_regularizer = kernel_regularizer
        self.wq = self.wk = self.wv = None

    def get_config(self):
        return {'activation': self.activation, 'name': self.name}

    def build(self, F, E, dev):
        self.wq = _glorot((F, E), dev, self._gen)
        self.wk = _glorot((E, E), dev, self._gen)
        self.wv = _glorot((F, F), dev, self._gen)
        self.built = True

    def __call__(self, inputs):
        nodes, nlist, edges, inv_degree = inputs
        dev = _device()
        nodes, nlist = _f32(nodes, dev), _i32(nlist, dev)
        edges, inv = _f32(edges, dev), _f32(inv_degree, dev).reshape(-1)
        N, F = nodes.shape
        K, E = nlist.shape[1], edges.shape[-1]
        if not self.built:
            self.build(F, E, dev)
        agg = torch.empty(N, F, dtype=torch.float32, device=dev)
        ctx = self._ctx(dev)
        ctx.check(ctx.lib.ng_amp_attend(ctx.handle, self._st(dev), N, K, F, E, ptr(nodes), ptr(nlist),
                                        ptr(edges), ptr(inv), ptr(self.wq.contiguous()),
                                        ptr(self.wk.contiguous()), ptr(agg)), "ng_amp_attend")
        zero = torch.zeros(F, dtype=torch.float32, device=dev)
        out = self._dense(agg, self.wv, zero, self.activation, False, dev)
        self._saved = (nodes, nlist, edges, inv, agg, out)
        return out

    def backward(self, dout):
        """Gradients of the last call: returns ``(dnodes [N,F], dedges [N,K,E])`` and leaves the weight gradients in
        ``self.grads`` (keys 'wq', 'wk', 'wv').  The sums run in a fixed order (incoming-slot lists, no atomics)."""
        if getattr(self, "_saved", None) is None:
            raise RuntimeError("AMPLayer.backward needs a forward call first")
        nodes, nlist, edges, inv, agg, out = self._saved
        dev = nodes.device
        N, F = nodes.shape
        K, E = nlist.shape[1], edges.shape[-1]
        dagg, dwv, _ = self._dense_bwd(agg, self.wv, out, _f32(dout, dev).reshape(N, F), self.activation, dev)
        tgt = nlist.reshape(-1).to(torch.int64)
        in_slot = torch.argsort(tgt, stable=True).to(torch.int32).contiguous()
        in_ptr = torch.zeros(N + 1, dtype=torch.int64, device=dev)
        in_ptr[1:] = torch.cumsum(torch.bincount(tgt, minlength=N), 0)
        in_ptr = in_ptr.to(torch.int32).contiguous()
        dh = torch.empty(N, F, dtype=torch.float32, device=dev)
        de = torch.empty(N, K, E, dtype=torch.float32, device=dev)
        dwq = torch.empty(F, E, dtype=torch.float32, device=dev)
        dwk = torch.empty(E, E, dtype=torch.float32, device=dev)
        ctx = self._ctx(dev)
        ctx.check(ctx.lib.ng_amp_attend_bwd(ctx.handle, self._st(dev), N, K, F, E, ptr(nodes), ptr(nlist), ptr(edges),
                                            ptr(inv), ptr(self.wq.contiguous()), ptr(self.wk.contiguous()),
                                            ptr(in_ptr), ptr(in_slot), ptr(dagg), ptr(dh), ptr(de), ptr(dwq),
                                            ptr(dwk)), "ng_amp_attend_bwd")
        self.grads = {"wq": dwq, "wk": dwk, "wv": dwv}
        return dh, de
