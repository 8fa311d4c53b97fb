"""Structure front end: PDB (.pdb / .pdb.gz, multi-MODEL) reader and the k-nearest-neighbour graph
builder behind ``universe2graph`` (nmrgnn/library.py:106-117; the arithmetic lives in the external
``nmrdata.parse_universe`` which is not in the tree -> conventions below are ours, parity unpinned):

  * coordinates are read in Angstrom and distances are reported in nm (x 0.1): the model's RBF grid
    spans 0.005-0.2 (model.py:31-32) and library.py:107 says "Universe is presumed to be in Angstrom";
  * neighbours = the K nearest OTHER atoms, ascending distance; fewer than K atoms -> trailing slots
    padded with nlist = 0, edges = 0 (the mask convention of model.py:251);
  * atoms one-hot over nmrgnn_amd.standards.ELEMENTS; unknown elements map to 'X'.
"""
from __future__ import annotations

import gzip

import numpy as np

from .standards import ELEMENTS, load_embeddings


class Structure:
    """minimal stand-in for the attributes of an MDAnalysis Universe that the path reads"""

    def __init__(self, names, resnames, resids, elements, frames):
        self.names = np.asarray(names)
        self.resnames = np.asarray(resnames)
        self.resids = np.asarray(resids)
        self.elements = np.asarray(elements)
        self.frames = [np.asarray(f, dtype=np.float32) for f in frames]   # list of [N,3] Angstrom
        self.frame = 0

    @property
    def n_atoms(self):
        return len(self.names)

    @property
    def positions(self):
        return self.frames[self.frame]

    def __len__(self):
        return len(self.frames)

    def trajectory(self):
        for i in range(len(self.frames)):
            self.frame = i
            yield i


def _element_from(line, name):
    el = line[76:78].strip() if len(line) >= 78 else ""
    if not el:
        el = "".join(c for c in name if c.isalpha())[:1]
    el = el.capitalize()
    return el


def read_pdb(path):
    """ATOM/HETATM records; every MODEL becomes a frame (atom order/identity taken from the first)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    names, resnames, resids, elements = [], [], [], []
    frames, cur = [], []
    first_done = False
    with opener(path, "rt") as f:
        for line in f:
            rec = line[:6]
            if rec in ("ATOM  ", "HETATM"):
                cur.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
                if not first_done:
                    name = line[12:16].strip()
                    names.append(name)
                    resnames.append(line[17:20].strip())
                    try:
                        resids.append(int(line[22:26]))
                    except ValueError:
                        resids.append(0)
                    elements.append(_element_from(line, name))
            elif rec.startswith("ENDMDL"):
                if cur:
                    frames.append(cur)
                    cur = []
                    first_done = True
    if cur:
        frames.append(cur)
    n = len(names)
    frames = [np.asarray(fr, np.float32) for fr in frames if len(fr) == n]
    if not frames:
        raise ValueError(f"no atoms found in {path}")
    return Structure(names, resnames, resids, elements, frames)


def knn_graph(positions, K=16, scale=0.1):
    """(nlist[N,K] int32, edges[N,K] f32): K nearest other atoms, ascending; distances * scale."""
    from scipy.spatial import cKDTree
    pos = np.asarray(positions, np.float64)
    N = pos.shape[0]
    kq = min(K + 1, N)
    tree = cKDTree(pos)
    dist, idx = tree.query(pos, k=kq)
    if kq == 1:
        dist, idx = dist[:, None], idx[:, None]
    nlist = np.zeros((N, K), np.int32)
    edges = np.zeros((N, K), np.float32)
    # drop self (normally column 0; with coincident atoms it may sit elsewhere)
    rows = np.arange(N)
    self_col = np.argmax(idx == rows[:, None], axis=1)
    has_self = (idx == rows[:, None]).any(axis=1)
    keep = np.ones_like(idx, dtype=bool)
    keep[rows[has_self], self_col[has_self]] = False
    keep[~has_self, -1] = False
    kk = kq - 1
    nl = idx[keep].reshape(N, kk)
    dd = dist[keep].reshape(N, kk)
    nlist[:, :kk] = nl
    edges[:, :kk] = (dd * scale).astype(np.float32)
    return nlist, edges


def atoms_onehot(elements):
    table = load_embeddings()['atom']
    idx = np.array([table.get(e, table['X']) for e in elements], np.int64)
    out = np.zeros((len(idx), len(ELEMENTS)), np.float32)
    out[np.arange(len(idx)), idx] = 1.0
    return out


def inv_degree_of(nlist):
    """nmrgnn/library.py:115-116: 1 / #(nlist > 0), 0 when that count is 0 (index 0 never counts)."""
    deg = (np.asarray(nlist) > 0).sum(axis=1).astype(np.float32)
    out = np.zeros_like(deg)
    np.divide(1.0, deg, out=out, where=deg > 0)
    return out
