"""Drop-in library surface: PDB -> graph -> model(g) -> check_peaks (reference README.md:75-106,
tests/test_nmrgnn.py:227-257).  The PDB files under tests/data are the reference's own fixtures."""
import os
import warnings

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PDB1 = os.path.join(HERE, "data", "108M.pdb")
PDB2 = os.path.join(HERE, "data", "7lgi.pdb.gz")


def test_universe2graph_conventions():
    import nmrgnn_amd
    from nmrgnn_amd.structure import read_pdb
    s = read_pdb(PDB1)
    assert s.n_atoms == 2482 and len(s) == 1
    atoms, nlist, edges, inv = nmrgnn_amd.universe2graph(s)
    assert atoms.shape == (2482, 10) and nlist.shape == (2482, 16) and edges.shape == (2482, 16)
    assert inv.shape == (2482,) and atoms.dtype == np.float32 and edges.dtype == np.float32
    assert np.all(atoms.sum(1) == 1)
    # brute-force check of a few rows: K nearest others, ascending, nm units
    pos = s.positions.astype(np.float64)
    for i in (0, 77, 1234, 2481):
        d = np.linalg.norm(pos - pos[i], axis=1)
        d[i] = np.inf
        order = np.argsort(d, kind="stable")[:16]
        np.testing.assert_allclose(edges[i], d[order] * 0.1, rtol=1e-5)
        assert set(nlist[i]) == set(order)
    assert np.all(np.diff(edges, axis=1) >= -1e-7)
    deg = (nlist > 0).sum(1)
    np.testing.assert_allclose(inv, 1.0 / deg)
    assert (nlist == 0).sum() == 14          # real neighbours with index 0
    # the same through a path and through a Universe-like object
    a2 = nmrgnn_amd.universe2graph(PDB1)[0]
    np.testing.assert_array_equal(a2, atoms)

    class FakeAtoms:
        positions = s.positions
        elements = s.elements
        names = s.names

    class FakeUniverse:
        atoms = FakeAtoms()
    np.testing.assert_array_equal(nmrgnn_amd.universe2graph(FakeUniverse())[1], nlist)


def test_multimodel_gz_and_small_structures():
    from nmrgnn_amd.structure import knn_graph, read_pdb
    s = read_pdb(PDB2)
    assert s.n_atoms == 2770 and len(s) == 10
    frames = [f for f in s.trajectory()]
    assert frames == list(range(10))
    nl, e = knn_graph(np.array([[0., 0, 0], [1, 0, 0], [0, 2, 0]]), K=4)
    np.testing.assert_array_equal(nl, [[1, 2, 0, 0], [0, 2, 0, 0], [0, 1, 0, 0]])
    np.testing.assert_allclose(e[0], [0.1, 0.2, 0, 0], rtol=1e-6)
    nl, e = knn_graph(np.zeros((1, 3)), K=3)
    assert np.all(nl == 0) and np.all(e == 0)


def test_check_peaks_matches_oracle_semantics():
    import nmrgnn_amd
    from nmrgnn_amd.standards import load_standards
    from oracle import nmrgnn_oracle as O
    rng = np.random.default_rng(0)
    elem = rng.choice([2, 3, 4, 5], size=200, p=[0.4, 0.1, 0.45, 0.05])
    atoms = np.eye(10, dtype=np.float32)[elem]
    st = load_standards()
    peaks = np.array([st[e][1] + rng.normal() * 1.5 * max(st[e][2], 1.0) for e in elem])
    np.testing.assert_array_equal(nmrgnn_amd.check_peaks(atoms, peaks), O.check_peaks(atoms, peaks, st))
    with pytest.raises(Warning):
        nmrgnn_amd.check_peaks(atoms, peaks + 1e4)


@pytest.mark.gpu
def test_end_to_end_108M_and_trajectory(gpu_device, tmp_path):
    """config #1 / #5 plumbing: whole-protein graphs (N=2482 / 2770, variable index locality) through
    the baseline architecture (F=256) against the oracle under the same seeded weights."""
    import nmrgnn_amd
    from nmrgnn_amd.structure import read_pdb
    from oracle import nmrgnn_oracle as O
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        model = nmrgnn_amd.load_model()
    g = nmrgnn_amd.universe2graph(PDB1)
    peaks = model(g)
    assert peaks.shape == (2482,) and hasattr(peaks, "numpy")
    sd = model.get_weights()
    ohp = O.hypers(**model.hypers.as_dict())
    ref = O.gnn_forward(g, sd, ohp, model.peak_std[:10], model.peak_avg[:10])
    # The head output is multiplied by the real peak_std (N: 50.94): 1e-4 on the STANDARDISED prediction is the north
    # star's budget at std = 1; on the de-standardised N shifts float32 itself stops at ~6e-4 — the reference's own
    # traced graph evaluated in float32 sits 5.7e-4 from its float64 value on this protein (tests/test_gpu_savedmodel.py
    # holds the per-element comparison against the reference graph; measured there: C 1.7e-4, N 6.3e-4, H 1.0e-4).
    err = np.abs(np.asarray(peaks) - ref)
    std = np.asarray(model.peak_std[:10])[np.argmax(g[0], axis=1)]
    assert np.max(err[std > 0] / std[std > 0]) < 5e-5
    assert np.max(err[std == 0]) == 0.0
    conf = None
    try:
        conf = nmrgnn_amd.check_peaks(g[0], peaks)
    except Warning:
        pass                                                    # random weights may look "awful"
    # save / reload round trip
    model.save(str(tmp_path / "m"))
    m2 = nmrgnn_amd.load_model(str(tmp_path / "m"))
    np.testing.assert_array_equal(np.asarray(m2(g)), np.asarray(peaks))
    # the same directory read as a Keras SavedModel: only variables/variables.{index,data-*} (TF bundle)
    import os
    os.remove(tmp_path / "m" / "config.json")
    os.remove(tmp_path / "m" / "weights.npz")
    m3 = nmrgnn_amd.load_model(str(tmp_path / "m"))
    assert m3.hypers.get('atom_feature_size') == 256
    np.testing.assert_array_equal(np.asarray(m3(g)), np.asarray(peaks))
    # trajectory: frames differ
    s = read_pdb(PDB2)
    first = None
    for _ in s.trajectory():
        pk = np.asarray(model(nmrgnn_amd.universe2graph(s)))
        assert pk.shape == (2770,)
        first = pk if first is None else first
    assert np.mean((pk - first) ** 2) > 0


def test_eval_struct_requires_a_structure():
    from nmrgnn_amd.main import eval_structure
    with pytest.raises(ValueError, match="at least"):          # nmrgnn/main.py:201-202
        eval_structure((), "out.csv")


def test_eval_struct_cli_surface():
    from click.testing import CliRunner
    from nmrgnn_amd.main import main
    res = CliRunner().invoke(main, ["eval-struct", "--help"])
    assert res.exit_code == 0
    for opt in ("--model-file", "--neighbor-number", "--stride", "STRUCT_FILES", "OUTPUT_CSV"):
        assert opt in res.output


@pytest.mark.gpu
def test_eval_struct_trajectory_csv(gpu_device, tmp_path):
    """config #5 driver: 10-model trajectory, stride 3, batched frames == frame-by-frame predictions."""
    import csv
    import nmrgnn_amd
    from nmrgnn_amd.main import eval_structure
    from nmrgnn_amd.structure import read_pdb
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        timing = eval_structure((PDB2,), str(tmp_path / "o.csv"), stride=3, frames_per_batch=3,
                                keep_going=True, echo=lambda *_: None)
        model = nmrgnn_amd.load_model()
    assert set(timing) == {'Structure', 'Model Inference (MI355X)', 'Parsing'}
    with open(tmp_path / "o.csv") as f:
        rows = list(csv.reader(f))
    assert rows[0] == ['index', 'residues', 'resids', 'names', 'peaks', 'confident', 'time', 'frame']
    assert len(rows) == 1 + 4 * 2770                              # frames 0,3,6,9
    assert [r[7] for r in rows[1::2770]] == ['0', '3', '6', '9']
    s = read_pdb(PDB2)
    s.frame = 9
    ref = np.round(np.asarray(model(nmrgnn_amd.universe2graph(s))).astype(np.float64), 2)
    got = np.array([float(r[4]) for r in rows[1 + 3 * 2770:]])
    assert np.max(np.abs(got - ref)) <= 0.011                     # rounding boundary only
    assert rows[1][1] == s.resnames[0] and rows[1][3] == s.names[0]


@pytest.mark.gpu
@pytest.mark.parametrize("K", [16, 5, 24])
def test_gpu_knn_matches_host_builder(gpu_device, K):
    """ng_knn_graph vs the host cKDTree builder on the reference's PDB fixtures: same neighbours in the
    same order (up to exact distance ties), distances to 1 ulp-ish, same inv_degree, frame offsets."""
    from nmrgnn_amd.graph import frames_to_batch
    from nmrgnn_amd.structure import atoms_onehot, inv_degree_of, knn_graph, read_pdb
    s = read_pdb(PDB2)
    frames = np.stack(s.frames[:3])
    atoms = atoms_onehot(s.elements)
    gb = frames_to_batch(atoms, frames, K, device=gpu_device)
    n = atoms.shape[0]
    assert gb.N == 3 * n and gb.G == 3 and gb.K == K
    nl, ed, inv = gb.nlist.cpu().numpy(), gb.edges.cpu().numpy(), gb.inv_degree.cpu().numpy()
    for f in range(3):
        hn, he = knn_graph(frames[f], K)
        sl = slice(f * n, (f + 1) * n)
        np.testing.assert_allclose(ed[sl], he, rtol=2e-6, atol=1e-7)
        same = (nl[sl] - f * n) == hn
        tied = np.isclose(he, np.roll(he, 1, axis=1), rtol=1e-6) | np.isclose(he, np.roll(he, -1, axis=1), rtol=1e-6)
        assert np.all(same | tied)
        assert same.mean() > 0.999
        np.testing.assert_allclose(inv[sl], inv_degree_of(nl[sl] - f * n), rtol=1e-6)


@pytest.mark.gpu
def test_gpu_knn_tiny_and_ragged(gpu_device):
    from nmrgnn_amd.graph import frames_to_batch
    pos = np.array([[[0, 0, 0], [1, 0, 0], [0, 2, 0]]], np.float32)      # n-1 < K -> padded slots
    atoms = np.eye(10, dtype=np.float32)[[4, 2, 3]]
    gb = frames_to_batch(atoms, pos, 16, device=gpu_device)
    nl, ed, inv = gb.nlist.cpu().numpy(), gb.edges.cpu().numpy(), gb.inv_degree.cpu().numpy()
    assert nl[0, :2].tolist() == [1, 2] and np.allclose(ed[0, :2], [0.1, 0.2]) and not ed[0, 2:].any()
    assert nl[1, :2].tolist() == [0, 2] and not nl[:, 2:].any()
    np.testing.assert_allclose(inv, [0.5, 1.0, 1.0])                     # index 0 never counts (library.py:115)


@pytest.mark.gpu
@pytest.mark.parametrize("n,K", [(1, 16), (7, 16), (33, 16), (1500, 16), (3000, 16), (70, 40), (2100, 40), (65, 64)])
def test_gpu_knn_kernels_give_identical_lists(gpu_device, monkeypatch, n, K):
    """the one-wave-per-query kernel (default up to 16384 queries of frames up to 4096 atoms), the 8 / 16-lanes-per-query kernels
    (NG_KNN=lanes) and the one-lane kernel (NG_KNN=serial) give identical lists, exact distance ties included (integer grid coordinates): (distance, index)
    ascending in all of them; three frames, so that the frame offsets take part"""
    from nmrgnn_amd.graph import frames_to_batch
    rng = np.random.default_rng(n + K)
    pos = rng.integers(0, 6 if n < 2000 else 14, size=(3, n, 3)).astype(np.float32)
    atoms = np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
    out = {}
    for mode in ("serial", "lanes", "wave"):
        if mode == "wave":
            monkeypatch.delenv("NG_KNN", raising=False)
        else:
            monkeypatch.setenv("NG_KNN", mode)
        gb = frames_to_batch(atoms, pos, K, device=gpu_device)
        out[mode] = (gb.nlist.cpu().numpy(), gb.edges.cpu().numpy(), gb.inv_degree.cpu().numpy())
    for mode in ("lanes", "wave"):
        for a, b in zip(out["serial"], out[mode]):
            assert np.array_equal(a, b), mode


@pytest.mark.gpu
def test_gpu_knn_many_queries_default_path(gpu_device, monkeypatch):
    """more than 16384 queries in a call: the default goes back to the lanes-per-query kernels (the wave-per-query kernel is for
    molecule-sized calls); still the serial kernel's lists"""
    from nmrgnn_amd.graph import frames_to_batch
    rng = np.random.default_rng(5)
    n, G = 1100, 64
    pos = (rng.random((G, n, 3)) * 30).astype(np.float32)
    atoms = np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
    out = {}
    for mode in ("serial", "wave"):
        if mode == "wave":
            monkeypatch.delenv("NG_KNN", raising=False)
        else:
            monkeypatch.setenv("NG_KNN", mode)
        gb = frames_to_batch(atoms, pos, 16, device=gpu_device)
        out[mode] = (gb.nlist.cpu().numpy(), gb.edges.cpu().numpy(), gb.inv_degree.cpu().numpy())
    for a, b in zip(out["serial"], out["wave"]):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_save_and_load_resume_training_with_adam_state(gpu_device, tmp_path):
    """model.save keeps the Adam slots and step count (as the reference's checkpoints do, main.py:63-68): three steps ==
    two steps + save + load into a fresh model + one step, bit for bit (explicit seeds)"""
    import torch
    import nmrgnn_amd
    from nmrgnn_amd import synth
    from nmrgnn_amd.graph import GraphBatch
    from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
    from nmrgnn_amd.model import GNNModel
    from nmrgnn_amd.standards import load_standards
    from nmrgnn_amd.train import Trainer
    hp = declare_gnn_space(HyperParameters(atom_feature_size=64))
    b = synth.make_batch(6, 40, 16, 10, 0.1, seed=3)

    def fresh():
        m = GNNModel(hp, load_standards(), device=gpu_device, seed=5)
        m.build(10)
        return m

    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    y, w = torch.from_numpy(b["y"]).to(gpu_device), torch.from_numpy(b["w"]).to(gpu_device)
    m1 = fresh()
    t1 = Trainer(m1.engine, lr=1e-3)
    for s in range(3):
        t1.step(gb, y, w, seed=100 + s)
    ref = m1.engine.params.flat.clone()
    m2 = fresh()
    t2 = Trainer(m2.engine, lr=1e-3)
    for s in range(2):
        t2.step(gb, y, w, seed=100 + s)
    m2.save(str(tmp_path / "ckpt"))
    m3 = nmrgnn_amd.load_model(str(tmp_path / "ckpt"), device=gpu_device)
    assert m3.engine.adam_t == 2
    t3 = Trainer(m3.engine, lr=1e-3)
    t3.step(gb, y, w, seed=102)
    assert torch.equal(m3.engine.params.flat, ref)
