"""ng_knn_graph through the cell grid (knn_cells.hip, frames of >= 16384 atoms by default; NG_KNN=cells forces it) against
the brute-force kernels (NG_KNN=brute): the SAME lists bit for bit — (distance, index) ascending, ties to the lower
index, self excluded, unused slots (0, 0.0), inv_degree as nmrgnn/library.py:115-116 — on uniform boxes, a protein tiled
to 55 k atoms, flat and linear point sets, duplicated positions, several frames, K up to 40, and a frame smaller than K."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(gpu_device, pos, K, scale=0.1):
    import torch
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    G, n, _ = pos.shape
    tp = torch.from_numpy(np.ascontiguousarray(pos, dtype=np.float32)).to(gpu_device)
    nl = torch.full((G * n, K), -7, dtype=torch.int32, device=gpu_device)
    ed = torch.full((G * n, K), -7.0, device=gpu_device)
    inv = torch.full((G * n,), -7.0, device=gpu_device)
    ctx = _lib.get_context(0)
    st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
    ctx.check(ctx.lib.ng_knn_graph(ctx.handle, st, G, n, K, scale, ptr(tp), ptr(nl), ptr(ed), ptr(inv)), "knn")
    torch.cuda.synchronize()
    return nl.cpu().numpy(), ed.cpu().numpy(), inv.cpu().numpy()


def _both(gpu_device, monkeypatch, pos, K):
    monkeypatch.setenv("NG_KNN", "brute")
    ref = _run(gpu_device, pos, K)
    monkeypatch.setenv("NG_KNN", "cells")
    got = _run(gpu_device, pos, K)
    for a, b, name in zip(got, ref, ("nlist", "edges", "inv_degree")):
        np.testing.assert_array_equal(a, b, err_msg=name)
    return ref


def _protein():
    from nmrgnn_amd.structure import read_pdb
    frames = read_pdb(os.path.join(HERE, "data", "7lgi.pdb.gz"))
    return np.asarray(frames.frames[0], dtype=np.float32)


@pytest.mark.parametrize("kind,n,K,G", [("box", 40000, 16, 1), ("box", 5000, 16, 3), ("box", 3000, 40, 2), ("box", 2000, 20, 1),
                                        ("plane", 20000, 16, 1), ("line", 4000, 16, 1), ("dup", 6000, 16, 1), ("tiny", 70, 16, 2),
                                        ("clusters", 30000, 16, 1)])
def test_cell_grid_lists_equal_brute_force(gpu_device, monkeypatch, kind, n, K, G):
    rng = np.random.default_rng(n + K)
    if kind == "box":
        pos = rng.random((G, n, 3)) * (n / 0.1) ** (1.0 / 3.0)            # ~0.1 atoms per cubic Angstrom
    elif kind == "plane":
        pos = rng.random((G, n, 3)) * 300.0
        pos[..., 2] = 4.25
    elif kind == "line":
        pos = np.zeros((G, n, 3))
        pos[..., 0] = rng.random((G, n)) * 2000.0
    elif kind == "dup":                                                    # every position three times: ties by index
        base = rng.random((G, n // 3, 3)) * 40.0
        pos = np.concatenate([base, base, base], axis=1)
    elif kind == "tiny":
        pos = rng.random((G, n, 3)) * 12.0
    else:                                                                  # dense blobs far apart: most cells empty
        centres = rng.random((30, 3)) * 2000.0
        pos = (centres[rng.integers(0, 30, n)] + rng.standard_normal((n, 3)) * 6.0)[None]
    nl, ed, inv = _both(gpu_device, monkeypatch, pos.astype(np.float32), K)
    # and the lists are right: distances ascending, first neighbour checked against a float64 search on a sample
    assert (np.diff(ed, axis=1) >= 0).all()
    p64 = pos.astype(np.float32).astype(np.float64)
    for g in range(G):
        for i in rng.integers(0, n, 20):
            d = np.sqrt(((p64[g] - p64[g, i]) ** 2).sum(-1))
            d[i] = np.inf
            assert abs(ed[g * n + i, 0] - 0.1 * d.min()) < 1e-5 * max(1.0, 0.1 * d.min())


def test_cell_grid_on_a_tiled_protein_and_by_default_at_that_size(gpu_device, monkeypatch):
    """7lgi (2770 atoms) tiled 4 x 5 to 55,400 atoms: protein density inside, empty space between the copies.  At this size
    the cell grid is the DEFAULT path (no NG_KNN set)."""
    p = _protein()
    ext = p.max(0) - p.min(0) + 9.0
    tiles = [p + np.array([ix * ext[0], iy * ext[1], 0.0], dtype=np.float32) for ix in range(4) for iy in range(5)]
    pos = np.concatenate(tiles)[None]
    ref = _both(gpu_device, monkeypatch, pos, 16)
    monkeypatch.delenv("NG_KNN", raising=False)
    got = _run(gpu_device, pos, 16)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    # every copy sees the neighbours of the original, shifted by its offset
    n0 = p.shape[0]
    one = _run(gpu_device, p[None], 16)
    np.testing.assert_array_equal(got[0][:n0], one[0])
    np.testing.assert_allclose(got[1][:n0], one[1], rtol=0, atol=2e-6)
