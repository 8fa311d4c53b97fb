"""The oracle against the REFERENCE'S OWN TRACED GRAPH (CPU, no GPU needed).

tests/golden/golden_savedmodel.npz holds what the reference's SavedModel FunctionDefs
(`__inference__wrapped_model_4657940`, training=False, and
`__inference_gnn-model_layer_call_and_return_conditional_losses_4659631`, training=True — the traces of
nmrgnn/model.py:245-274) produce when interpreted op by op in float64 (tests/golden/make_savedmodel_exec.py,
which never touches the oracle).  The oracle must reproduce those numbers: this is the pin that ties
oracle/nmrgnn_oracle.py to arithmetic the reference itself defines.  The HIP path is held to the same
fixture in tests/test_gpu_savedmodel.py.
"""
import numpy as np
import pytest

from helpers import SAVEDMODEL_CASES, load_savedmodel_case
from oracle import nmrgnn_oracle as O

# float64 oracle vs float64 execution of the reference graph: only re-association noise is allowed
F64_ATOL = 1e-9


@pytest.mark.parametrize("tag", SAVEDMODEL_CASES)
def test_oracle_equals_reference_graph_inference(tag):
    c = load_savedmodel_case(tag)
    hp = O.hypers(atom_feature_size=c["F"])
    out = O.gnn_forward((c["atoms"], c["nlist"], c["edges"], c["inv_degree"]), c["weights"], hp,
                        c["peak_std"], c["peak_avg"], return_all=True)
    assert np.max(np.abs(out["peaks"] - c["peaks64"])) < F64_ATOL
    if "e64" in c:
        # intermediates (stored rounded to float32): masked edge features, node features after the MP block
        assert np.max(np.abs(out["e"] - c["e64"])) < 1e-6
        assert np.max(np.abs(out["h"][::16] - c["h_mp64_rows16"])) < 2e-6
    # the fixture exercises what it should: padded slots, index-0 neighbours, elements with std = 0
    if tag == "padded":
        assert (c["edges"] == 0).any() and (c["inv_degree"] == 0).any()
        assert len(np.unique(np.argmax(c["atoms"], 1))) == 10
    assert ((c["nlist"] == 0) & (c["edges"] > 0)).any()
    assert np.ptp(c["peaks64"]) > 1.0


@pytest.mark.parametrize("tag", SAVEDMODEL_CASES)
def test_oracle_equals_reference_graph_training(tag):
    """training=True trace: GaussianNoise (sigma 0.025, mask from the un-noised distances) and Dropout
    (rate 0.2, kept units x1.25) with the SAME explicit draws fed to both."""
    c = load_savedmodel_case(tag)
    hp = O.hypers(atom_feature_size=c["F"])
    peaks = O.gnn_forward((c["atoms"], c["nlist"], c["edges"], c["inv_degree"]), c["weights"], hp,
                          c["peak_std"], c["peak_avg"], training=True, noise=c["train_xi"],
                          dropout_mask=c["train_keep"])
    assert np.max(np.abs(peaks - c["train_peaks64"])) < F64_ATOL
    assert np.max(np.abs(c["train_peaks64"] - c["peaks64"])) > 1e-2     # the draws really matter


def test_reference_fp32_execution_noise_is_recorded():
    """How far float32 evaluation of the reference's own graph sits from its float64 value — the
    yardstick for the 1e-4 budget once peak_std = 50.9 (N) multiplies the head output."""
    for tag, lo, hi in (("padded", 1e-6, 1e-3), ("pdb108m", 1e-5, 5e-3), ("lgi7", 1e-5, 5e-3), ("pdb108m_f64", 1e-5, 5e-3)):
        c = load_savedmodel_case(tag)
        err = np.max(np.abs(c["peaks32"].astype(np.float64) - c["peaks64"]))
        assert lo < err < hi, (tag, err)


def test_interpreter_cross_check_with_independent_contractions():
    """Insurance for the fixture generator (tests/golden/make_savedmodel_exec.py evaluates the graph's Einsum / MatMul nodes
    with np.einsum / @): the same FunctionDef with those ops evaluated by DIFFERENT code — tensordot with an explicit axis
    list, a hand-written loop over the edge-feature index, a transposed product — must give the committed numbers.  A
    shared misreading of an equation string would have to be made twice, in two notations.  Needs the reference tree
    (this container); skipped where /root/reference is absent."""
    import importlib.util
    import os
    pb = "/root/reference/nmrgnn/models/baseline/saved_model.pb"
    if not os.path.exists(pb):
        pytest.skip("reference SavedModel not available on this box")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_savedmodel_exec.py")
    spec = importlib.util.spec_from_file_location("make_savedmodel_exec", here)
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)

    class Other(M.Interp):
        def eval(self, n, value):
            op, a = n["op"], n["attr"]
            if op == "Einsum":
                x = [value(r) for r in n["input"] if not r.startswith("^")]
                eq = a["equation"]["s"]
                eq = eq.decode() if isinstance(eq, bytes) else eq
                self.ops_seen[op] = self.ops_seen.get(op, 0) + 1
                if eq == "lmn,ijl->mnij":       # w[l,m,n], gathered[i,j,l]
                    return np.tensordot(x[0], x[1], axes=([0], [2]))
                if eq == "mnij,ijn->mi":        # t[m,n,i,j], edges[i,j,n]: loop over n, reduce j last
                    t, e = x
                    out = np.zeros((t.shape[0], t.shape[2]), t.dtype)
                    for nn in range(t.shape[1]):
                        out += (t[:, nn, :, :] * e[None, :, :, nn]).sum(axis=2)
                    return out
                if eq == "mi,i->im":
                    return (x[0] * x[1][None, :]).T.copy()
                raise AssertionError(f"unexpected einsum {eq}")
            if op == "MatMul":
                x = [value(r) for r in n["input"] if not r.startswith("^")]
                self.ops_seen[op] = self.ops_seen.get(op, 0) + 1
                A = x[0].T if a.get("transpose_a", {}).get("b", False) else x[0]
                B = x[1].T if a.get("transpose_b", {}).get("b", False) else x[1]
                return np.dot(B.T, A.T).T
            return super().eval(n, value)

    funcs, top = M.load_functions()
    c = load_savedmodel_case("padded")
    g = M.graph_padded()
    w = M.seeded_weights(64, 4658)
    assert M.weights_digest(w) == str(c["weights_sha256"]) if "weights_sha256" in c else True
    fd = funcs[M.FN_INFER]
    vmap = M.variable_map(fd)
    variables = {res: w[key] for res, key in vmap.items()}
    centers = next(v for v in top.values() if v.shape == (128,))
    gap = next(v for v in top.values() if v.shape == () and v.dtype == np.float32 and 1e-3 < v < 2e-3)
    names = [nm for nm, dt in fd["args"] if dt != M.DT_RESOURCE]
    feeds = dict(zip(names, [g[0], g[1], g[2], g[3], centers, gap]))
    for ft, tol in ((np.float64, 1e-9), (np.float32, 2e-4)):
        it = Other(fd, ft, variables)
        out = np.asarray(it.run(feeds), np.float64)
        assert it.ops_seen.get("Einsum", 0) == 12 and it.ops_seen.get("MatMul", 0) >= 9
        assert np.max(np.abs(out - c["peaks64"])) < tol, ft
