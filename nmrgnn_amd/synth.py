"""Synthetic molecular-graph batches of the shape BASELINE.json names:
G graphs x n atoms, K neighbours drawn without replacement from the same graph (self excluded),
a fraction of trailing slots zero-padded (nlist = 0, edges = 0), distances ~ U(0.09, 0.45) (nm-scale,
the 16-NN range of the reference's PDB fixtures / 10), elements ~ Categorical over {C, N, H} =
indices {2, 3, 4} with the 108M.pdb mix, inv_degree exactly as nmrgnn/library.py:115-116
(counts nlist > 0).  Host-side numpy; no reference code involved."""
from __future__ import annotations

import numpy as np


def make_graph(n_atoms=256, K=16, num_elem=10, p_pad=0.05, rng=None, d_lo=0.09, d_hi=0.45):
    rng = np.random.default_rng(0) if rng is None else rng
    K_eff = min(K, n_atoms - 1)
    r = rng.random((n_atoms, n_atoms), dtype=np.float32)
    np.fill_diagonal(r, np.inf)
    nl = np.argpartition(r, K_eff - 1, axis=1)[:, :K_eff].astype(np.int64)
    if K_eff < K:
        nl = np.concatenate([nl, np.zeros((n_atoms, K - K_eff), np.int64)], axis=1)
    # at least one real edge to local atom 0 (exercises the nlist>0 degree quirk)
    if n_atoms > 1 and not np.any(nl[1, :K_eff] == 0):
        nl[1, 0] = 0
    n_pad = rng.binomial(K, p_pad, size=n_atoms)
    n_pad = np.maximum(n_pad, K - K_eff)
    slot = np.arange(K)[None, :]
    real = slot < (K - n_pad)[:, None]
    d = rng.uniform(d_lo, d_hi, size=(n_atoms, K)).astype(np.float32)
    nl = np.where(real, nl, 0)
    d = np.where(real, d, 0.0).astype(np.float32)
    elem = rng.choice([2, 3, 4], size=n_atoms, p=[0.35, 0.10, 0.55])
    atoms = np.zeros((n_atoms, num_elem), np.float32)
    atoms[np.arange(n_atoms), elem] = 1.0
    return atoms, nl, d


def inv_degree(nlist):
    """nmrgnn/library.py:115-116: divide_no_nan(1, sum(nlist > 0))"""
    deg = (nlist > 0).sum(axis=1).astype(np.float32)
    out = np.zeros_like(deg)
    np.divide(1.0, deg, out=out, where=deg > 0)
    return out


def make_batch(n_graphs=512, n_atoms=256, K=16, num_elem=10, p_pad=0.05, seed=42):
    """Returns dict with the concatenated tuple (global neighbour indices), graph_ptr and labels."""
    rng = np.random.default_rng(seed)
    atoms, nlist, edges, inv = [], [], [], []
    ptr = [0]
    off = 0
    for _ in range(n_graphs):
        a, nl, d = make_graph(n_atoms, K, num_elem, p_pad, rng)
        # degree is computed on the graph-LOCAL list, as the reference does per graph
        inv.append(inv_degree(nl))
        atoms.append(a)
        nlist.append(nl + off)
        edges.append(d)
        off += n_atoms
        ptr.append(off)
    N = off
    y = rng.standard_normal(N).astype(np.float32)
    w = np.ones(N, np.float32)
    return dict(atoms=np.concatenate(atoms), nlist=np.concatenate(nlist).astype(np.int32),
                edges=np.concatenate(edges), inv_degree=np.concatenate(inv),
                graph_ptr=np.asarray(ptr, np.int32), y=y, w=w)
