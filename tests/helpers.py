"""shared helpers for the parity tests (test infrastructure)"""
import numpy as np

from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
from nmrgnn_amd import synth


def make_hp(**kw):
    hp = HyperParameters(**kw)
    declare_gnn_space(hp)
    return hp


def hp_to_oracle(hp):
    from oracle import nmrgnn_oracle as O
    return O.hypers(**hp.as_dict())


def small_batch(n_graphs=3, n_atoms=50, K=16, num_elem=10, seed=7, p_pad=0.1):
    return synth.make_batch(n_graphs, n_atoms, K, num_elem, p_pad, seed)


def randomize_biases(engine, seed=3, scale=0.1):
    """non-zero biases so the bias / mask paths are exercised"""
    rng = np.random.default_rng(seed)
    sd = engine.params.state_dict()
    for k in sd:
        if k.endswith("bias"):
            sd[k] = (scale * rng.standard_normal(sd[k].shape)).astype(np.float32)
    engine.params.load_state_dict(sd)
    return sd


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


# ---- reference-graph fixture (tests/golden/golden_savedmodel.npz, made by make_savedmodel_exec.py) ----
def savedmodel_weight_shapes(F, E=3, H=128, C=10):
    out = []
    for t in range(4):
        ko = H if t < 3 else E
        out += [(f"edge_fc/{t}/kernel", (H, ko)), (f"edge_fc/{t}/bias", (ko,))]
    out += [(f"mp/{l}/w", (F, F, E)) for l in range(4)]
    for t in range(4):
        ko = F if t < 3 else F // 2
        out += [(f"fc/{t}/kernel", (F, ko)), (f"fc/{t}/bias", (ko,))]
    out += [("out/kernel", (F // 2, C)), ("out/bias", (C,)), ("embed/kernel", (C, F))]
    return out


def savedmodel_seeded_weights(F, seed, bias_scale=0.05):
    """the weight generator of tests/golden/make_savedmodel_exec.py (same draws; pinned by SHA-256)"""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in savedmodel_weight_shapes(F):
        if len(shape) == 1:
            p[name] = (bias_scale * rng.standard_normal(shape)).astype(np.float32)
            continue
        if len(shape) == 2:
            fi, fo = shape
        else:
            fi, fo = shape[1] * shape[0], shape[2] * shape[0]
        lim = np.sqrt(6.0 / (fi + fo))
        p[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
    return p


def load_savedmodel_case(tag):
    """One case of the reference-executed fixture: dict with the input tuple, seeded weights (digest
    checked), explicit training draws and the outputs of the reference's traced graph."""
    import hashlib
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_savedmodel.npz"))
    F = int(z[f"{tag}:F"])
    w = savedmodel_seeded_weights(F, int(z[f"{tag}:weight_seed"]))
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k], np.float32).tobytes())
    assert h.hexdigest() == str(z[f"{tag}:weights_sha256"]), "seeded weights differ from the fixture's"
    gt = str(z[f"{tag}:graph_of"]) if f"{tag}:graph_of" in z.files else tag      # a case on the graph of an earlier one
    elem = z[f"{gt}:elem"].astype(np.int64)
    atoms = np.eye(10, dtype=np.float32)[elem]
    edges = z[f"{gt}:edges"]
    N, K = edges.shape
    keep = np.unpackbits(z[f"{tag}:train_keep_bits"])[:N * (F // 2)].reshape(N, F // 2).astype(bool)
    c = dict(F=F, weights=w, atoms=atoms, nlist=z[f"{gt}:nlist"].astype(np.int32), edges=edges,
             inv_degree=z[f"{gt}:inv_degree"], peak_std=z["peak_std"], peak_avg=z["peak_avg"],
             peaks64=z[f"{tag}:peaks64"], peaks32=z[f"{tag}:peaks32"],
             train_xi=z[f"{tag}:train_xi"].astype(np.float32), train_keep=keep,      # (later cases store the draws as float16)
             train_peaks64=z[f"{tag}:train_peaks64"], train_peaks32=z[f"{tag}:train_peaks32"])
    if f"{tag}:e64" in z.files:      # intermediates of the float64 run: the first two cases only
        c.update(e64=z[f"{tag}:e64"], h_mp64_rows16=z[f"{tag}:h_mp64_rows16"])
    return c


SAVEDMODEL_CASES = ["padded", "pdb108m", "lgi7", "pdb108m_f64"]


# ---- full-batch oracle gradients, graph by graph (graphs are independent, so the batch gradient is the
#      sum of the per-graph gradients; each worker runs the float64 oracle on its share of graphs) ----
def _oracle_chunk(args):
    import os
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import nmrgnn_oracle as O
    (atoms, nlist, edges, inv, ptr, g0, g1, sd, ohp, dpe, std, avg, xi, mask) = args
    peaks = []
    total = None
    for g in range(g0, g1):
        a, b = int(ptr[g]), int(ptr[g + 1])
        pk, gr = O.gnn_forward_backward(
            (atoms[a:b], nlist[a:b] - a, edges[a:b], inv[a:b]), sd, ohp, dpe[a:b], std, avg,
            training=xi is not None, noise=None if xi is None else xi[a:b],
            dropout_mask=None if mask is None else mask[a:b])
        peaks.append(pk)
        if total is None:
            total = {k: np.array(v, np.float64) for k, v in gr.items()}
        else:
            for k, v in gr.items():
                total[k] += v
    return g0, np.concatenate(peaks), total


def oracle_batch_forward_backward(b, sd, ohp, dpeaks, std=None, avg=None, xi=None, mask=None, workers=None):
    """(peaks[N], {name: gradient}) of the float64 oracle over EVERY graph of batch dict ``b`` (needs
    graph_ptr; nlist holds global indices).  Spawned worker processes (no fork: the parent may hold a
    HIP context)."""
    import multiprocessing as mp
    import os
    ptr = np.asarray(b["graph_ptr"], np.int64)
    G = len(ptr) - 1
    if workers is None:
        workers = max(1, min(32, (os.cpu_count() or 1) // 2, G))
    per = -(-G // (workers * 4))
    jobs = []
    for g0 in range(0, G, per):
        g1 = min(G, g0 + per)
        a, z = int(ptr[g0]), int(ptr[g1])
        sub_ptr = ptr[g0:g1 + 1] - a
        jobs.append((b["atoms"][a:z], np.asarray(b["nlist"][a:z], np.int64) - a, b["edges"][a:z],
                     b["inv_degree"][a:z], sub_ptr, 0, g1 - g0, sd, ohp, np.asarray(dpeaks)[a:z], std, avg,
                     None if xi is None else xi[a:z], None if mask is None else mask[a:z]))
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        res = pool.map(_oracle_chunk, jobs)
    peaks = np.concatenate([r[1] for r in res])
    total = None
    for _, _, gr in res:
        if total is None:
            total = {k: v.copy() for k, v in gr.items()}
        else:
            for k, v in gr.items():
                total[k] += v
    return peaks, total
