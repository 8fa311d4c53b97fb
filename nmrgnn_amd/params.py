"""Flat parameter store: every trainable tensor of the model is a view into ONE contiguous fp32
device buffer (and its gradient into one flat gradient buffer), so that Adam is a single fused
kernel and the data-parallel gradient exchange is a single RCCL all-reduce.

Names / shapes / order follow the Keras variable tree of the reference's bundled model
(SURVEY App. A): edge-fc-block/dense_t {kernel,bias}, mp-block/MPLayer/w [F,F,E],
fc-block/dense_t {kernel,bias}, out_layer {kernel,bias}, embed_layer/kernel.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch


def param_shapes(hp, num_elem):
    F = hp.get('atom_feature_size')
    E = hp.get('edge_feature_size')
    H = hp.get('edge_hidden_size')
    Le, L, Lf = hp.get('edge_fc_layers'), hp.get('mp_layers'), hp.get('fc_layers')
    out = []
    for t in range(Le):                       # nmrgnn/model.py:119-128
        kout = H if t < Le - 1 else E
        out.append((f"edge_fc/{t}/kernel", (H, kout)))
        out.append((f"edge_fc/{t}/bias", (kout,)))
    for l in range(L):                        # nmrgnn/layers.py:11-18
        out.append((f"mp/{l}/w", (F, F, E)))
    for t in range(Lf):                       # nmrgnn/model.py:184-188
        kout = F if t < Lf - 1 else F // 2
        out.append((f"fc/{t}/kernel", (F, kout)))
        out.append((f"fc/{t}/bias", (kout,)))
    out.append(("out/kernel", (F // 2, num_elem)))   # nmrgnn/model.py:239
    out.append(("out/bias", (num_elem,)))
    out.append(("embed/kernel", (num_elem, F)))      # nmrgnn/model.py:241
    return out


def _glorot_limit(shape):
    # keras GlorotUniform: fans of a rank-3 weight are (shape[-2]*rf, shape[-1]*rf), rf = prod(shape[:-2])
    if len(shape) == 1:
        fi = fo = shape[0]
    elif len(shape) == 2:
        fi, fo = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fi, fo = shape[-2] * rf, shape[-1] * rf
    return math.sqrt(6.0 / (fi + fo))


class ParamStore:
    def __init__(self, hp, num_elem, device, seed=1234):
        self.shapes = OrderedDict(param_shapes(hp, num_elem))
        self.offsets = OrderedDict()
        off = 0
        for name, shape in self.shapes.items():
            off = (off + 3) // 4 * 4        # 16-byte aligned views (float4 loads in the kernels)
            self.offsets[name] = off
            off += int(np.prod(shape))
        self.numel = (off + 3) // 4 * 4
        self.device = device
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.views = OrderedDict()
        self.grad_views = OrderedDict()
        for name, shape in self.shapes.items():
            o, n = self.offsets[name], int(np.prod(shape))
            self.views[name] = self.flat[o:o + n].view(*shape)
            self.grad_views[name] = self.grad[o:o + n].view(*shape)
        self.on_change = None
        self.init_glorot(seed)

    def init_glorot(self, seed):
        """keras defaults: GlorotUniform kernels, zero biases (Dense; MPLayer.add_weight)."""
        rng = np.random.default_rng(seed)
        host = np.zeros(self.numel, dtype=np.float32)
        for name, shape in self.shapes.items():
            if name.endswith("bias"):
                continue
            lim = _glorot_limit(shape)
            o, n = self.offsets[name], int(np.prod(shape))
            host[o:o + n] = rng.uniform(-lim, lim, size=n).astype(np.float32)
        self.flat.copy_(torch.from_numpy(host))

    def __getitem__(self, name):
        return self.views[name]

    def g(self, name):
        return self.grad_views[name]

    def state_dict(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self.views.items()}

    def load_state_dict(self, sd):
        for k, v in sd.items():
            if k not in self.views:
                raise KeyError(f"unexpected parameter {k}")
            t = torch.as_tensor(np.asarray(v), dtype=torch.float32)
            if tuple(t.shape) != tuple(self.shapes[k]):
                raise ValueError(f"{k}: shape {tuple(t.shape)} != {tuple(self.shapes[k])}")
            self.views[k].copy_(t)
        if getattr(self, "on_change", None) is not None:
            self.on_change()              # packed weight images cached by the library are stale now

    def grads_dict(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self.grad_views.items()}

    def count(self):
        return sum(int(np.prod(s)) for s in self.shapes.values())
