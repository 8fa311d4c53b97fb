"""torch-CPU restatement of the reference forward in the op order TensorFlow executes.
TEST / BASELINE INFRASTRUCTURE ONLY — never imported by the product package.

Purpose
  (1) ``cpu_baseline`` leg of bench.py: the reference is TF/Keras and TensorFlow is not
      installed on either box, so the "reference TF-CPU path" is timed through this
      stand-in, which runs the same ops in the same order the traced SavedModel graph
      does (``lmn,ijl->mnij`` -> ``mnij,ijn->mi`` -> ``mi,i->im``,
      materialised gather and RBF tensors, one matmul+bias+softplus per Dense).
      kind = "port" in the bench line.
  (2) autograd cross-check of the hand-derived backward in nmrgnn_oracle.py.

Follows nmrgnn/model.py:245-274, nmrgnn/layers.py:26-46,137-140, nmrgnn/losses.py:30-39.
PARITY STATUS: parity unpinned (see nmrgnn_oracle.py header).
"""
from __future__ import annotations

import numpy as np
import torch

from . import nmrgnn_oracle as O


def to_torch_params(p, dtype=torch.float64, requires_grad=False):
    out = {}
    for k, v in p.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def _softplus(x):
    return torch.nn.functional.softplus(x, beta=1.0, threshold=30.0)


def mp_layer_ref_order(nodes, nlist, edges, inv_degree, w):
    """TF's pairwise lowering of einsum('ijn,ijl,lmn,i->im') [SURVEY App. A]."""
    sliced = nodes[nlist]                                   # GatherV2  [N,K,F]
    t = torch.einsum("lmn,ijl->mnij", w, sliced)            # [F,E,N,K] temporary
    t = torch.einsum("mnij,ijn->mi", t, edges)
    return torch.einsum("mi,i->im", t, inv_degree)


def mp_layer_alg_order(nodes, nlist, edges, inv_degree, w):
    sliced = nodes[nlist]
    A = torch.einsum("ijn,ijl->iln", edges, sliced)
    return inv_degree[:, None] * torch.einsum("iln,lmn->im", A, w)


def forward(inputs, p, hp, peak_std=None, peak_avg=None, training=False, noise=None,
            dropout_mask=None, order="ref"):
    atoms, nlist, d, inv = inputs
    dtype = p["embed/kernel"].dtype
    atoms = torch.as_tensor(atoms, dtype=dtype)
    nlist = torch.as_tensor(np.asarray(nlist), dtype=torch.int64)
    d = torch.as_tensor(d, dtype=dtype)
    inv = torch.as_tensor(inv, dtype=dtype)
    C = atoms.shape[-1]
    std = torch.ones(C, dtype=dtype) if peak_std is None else torch.as_tensor(peak_std, dtype=dtype)[:C]
    avg = torch.zeros(C, dtype=dtype) if peak_avg is None else torch.as_tensor(peak_avg, dtype=dtype)[:C]
    mask = (d > 0).to(dtype)[..., None]
    if training and hp["noise"] > 0:
        d = d + float(np.float32(hp["noise"])) * torch.as_tensor(noise, dtype=dtype)   # 0.025f in the traced graph
    centers, gap = O.rbf_centers(hp["rbf_low"], hp["rbf_high"], hp["edge_hidden_size"])
    centers = torch.tensor(centers, dtype=dtype)
    x = torch.exp(-(d[..., None] - centers) ** 2 / float(gap)) * mask
    Le = hp["edge_fc_layers"]
    for t in range(Le - 1):
        x = _softplus(x @ p[f"edge_fc/{t}/kernel"] + p[f"edge_fc/{t}/bias"])
    e = (x @ p[f"edge_fc/{Le-1}/kernel"] + p[f"edge_fc/{Le-1}/bias"]) * mask
    h = atoms @ p["embed/kernel"]
    mp = mp_layer_ref_order if order == "ref" else mp_layer_alg_order
    for l in range(hp["mp_layers"]):
        h = _softplus(mp(h, nlist, e, inv, p[f"mp/{l}/w"])) + h
    Lf = hp["fc_layers"]
    for t in range(Lf - 1):
        h = _softplus(h @ p[f"fc/{t}/kernel"] + p[f"fc/{t}/bias"]) + h
    g = _softplus(h @ p[f"fc/{Lf-1}/kernel"] + p[f"fc/{Lf-1}/bias"])
    if training and hp["dropout"]:
        g = g * torch.as_tensor(dropout_mask, dtype=dtype) / (1.0 - O.DROPOUT_RATE)
    full = g @ p["out/kernel"] + p["out/bias"]
    return torch.sum(full * atoms * std + atoms * avg, dim=-1)


def batch_loss_s1(y, w, pred, graph_ids, n_graphs):
    """mean over graphs of sum w (y-pred)^2 / sum w  (divide_no_nan)."""
    num = torch.zeros(n_graphs, dtype=pred.dtype).index_add_(0, graph_ids, w * (y - pred) ** 2)
    den = torch.zeros(n_graphs, dtype=pred.dtype).index_add_(0, graph_ids, w)
    per = torch.where(den != 0, num / torch.where(den != 0, den, torch.ones_like(den)),
                      torch.zeros_like(num))
    return per.sum() / n_graphs
