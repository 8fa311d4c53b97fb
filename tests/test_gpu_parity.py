"""GPU parity: the HIP path (through the C ABI) against the float64 oracle on the same inputs.
Tolerance: 1e-4 absolute on predicted shifts (BASELINE.json north_star), gradients 2e-4 relative
to the largest entry of each tensor."""
import numpy as np
import pytest

from helpers import make_hp, hp_to_oracle, small_batch, randomize_biases, rel_err

pytestmark = pytest.mark.gpu

PEAK_ATOL = 1e-4
GRAD_RTOL = 2e-4

CONFIGS = [
    dict(atom_feature_size=64, edge_feature_size=3, edge_hidden_size=128),   # bench arch
    dict(atom_feature_size=32, edge_feature_size=2, edge_hidden_size=32, mp_layers=2,
         fc_layers=3, edge_fc_layers=3),
    dict(atom_feature_size=256, edge_feature_size=3, edge_hidden_size=128),  # bundled-model arch
    dict(atom_feature_size=128, edge_feature_size=8, edge_hidden_size=64, mp_layers=1,
         fc_layers=2, edge_fc_layers=2),
    # the other choices of the reference's hyper-parameter space (nmrgnn/model.py:23,33-36)
    dict(atom_feature_size=64, edge_feature_size=3, edge_hidden_size=128, fc_activation='relu',
         mp_activation='tanh'),
    dict(atom_feature_size=128, edge_feature_size=8, edge_hidden_size=128, fc_activation='relu',
         mp_activation='relu', mp_layers=2),
    dict(atom_feature_size=32, edge_feature_size=64, edge_hidden_size=64, mp_layers=2, fc_layers=2,
         edge_fc_layers=3),
    dict(atom_feature_size=256, edge_feature_size=64, edge_hidden_size=128, mp_layers=1, fc_layers=2,
         edge_fc_layers=2, fc_activation='relu'),
]


def _setup(gpu_device, cfg, n_graphs=3, n_atoms=50, seed=7):
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    hp = make_hp(**cfg)
    b = small_batch(n_graphs, n_atoms, seed=seed)
    rng = np.random.default_rng(5)
    std = rng.uniform(0.5, 2.0, 10).astype(np.float32)
    avg = rng.uniform(-1.0, 1.0, 10).astype(np.float32)
    eng = Engine(hp, 10, std, avg, device=gpu_device, seed=11)
    sd = randomize_biases(eng)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"],
                    device=gpu_device)
    return hp, b, eng, sd, gb, std, avg


@pytest.mark.parametrize("cfg", CONFIGS)
def test_forward_inference_matches_oracle(gpu_device, cfg):
    import torch
    from oracle import nmrgnn_oracle as O
    hp, b, eng, sd, gb, std, avg = _setup(gpu_device, cfg)
    peaks = eng.forward(gb, training=False).cpu().numpy()
    ref = O.gnn_forward((b["atoms"], b["nlist"], b["edges"], b["inv_degree"]), sd, hp_to_oracle(hp),
                        std, avg)
    assert peaks.shape == ref.shape
    assert np.max(np.abs(peaks - ref)) < PEAK_ATOL, np.max(np.abs(peaks - ref))


@pytest.mark.parametrize("cfg", CONFIGS)
def test_training_forward_backward_matches_oracle(gpu_device, cfg):
    import torch
    from oracle import nmrgnn_oracle as O
    hp, b, eng, sd, gb, std, avg = _setup(gpu_device, cfg)
    N, K = b["edges"].shape
    Fh = hp.get('atom_feature_size') // 2
    xi = eng.randn(N * K, seed=123)
    mask = eng.dropout_mask(N * Fh, seed=321)
    peaks = eng.forward(gb, training=True, noise=xi, dropout_mask=mask)
    rng = np.random.default_rng(9)
    dpeaks = rng.standard_normal(N).astype(np.float32)
    eng.backward(torch.from_numpy(dpeaks).to(gpu_device))
    grads = eng.params.grads_dict()
    xi_h = xi.cpu().numpy().reshape(N, K)
    keep = 0.8
    mask_h = (mask.cpu().numpy().reshape(N, Fh) > 0).astype(np.float64)
    # sanity of the RNG kernels
    assert abs(xi_h.mean()) < 0.1 and abs(xi_h.std() - 1.0) < 0.1
    assert abs(mask_h.mean() - keep) < 0.05
    ref_peaks, ref_grads = O.gnn_forward_backward(
        (b["atoms"], b["nlist"], b["edges"], b["inv_degree"]), sd, hp_to_oracle(hp), dpeaks, std, avg,
        training=True, noise=xi_h, dropout_mask=mask_h)
    assert np.max(np.abs(peaks.cpu().numpy() - ref_peaks)) < PEAK_ATOL
    bad = {}
    for k, g in ref_grads.items():
        err = rel_err(grads[k], g)
        if err > GRAD_RTOL:
            bad[k] = err
    assert not bad, bad


def test_batch_invariance(gpu_device):
    """KAT-6: model(concat(g1,g2)) == concat(model(g1), model(g2))."""
    from nmrgnn_amd.graph import GraphBatch
    hp, b, eng, sd, gb, std, avg = _setup(gpu_device, CONFIGS[0], n_graphs=2, n_atoms=40)
    full = eng.forward(gb).cpu().numpy()
    n = 40
    for g in range(2):
        sl = slice(g * n, (g + 1) * n)
        sub = GraphBatch(b["atoms"][sl], b["nlist"][sl] - g * n, b["edges"][sl], b["inv_degree"][sl],
                         device=gpu_device)
        part = eng.forward(sub).cpu().numpy()
        np.testing.assert_allclose(part, full[sl], rtol=0, atol=2e-6)


@pytest.mark.parametrize("F,graphs,atoms", [(64, 48, 256), (256, 12, 200), (64, 512, 256)])      # the last: BASELINE configs[2] at full size
def test_relabelling_the_atoms_of_a_graph_permutes_the_peaks(gpu_device, F, graphs, atoms):
    """A property of the model that needs no oracle and holds at any size: the model sees atoms only through their lists, so
    relabelling the atoms of every graph (rows permuted, neighbour indices renamed, slot order kept) must permute the peaks.  Per
    atom every sum runs over the atom's own slots and features in an order that does not depend on where the atom sits in the batch
    (tiles, windows, workgroups), so the inference peaks must agree BIT FOR BIT; the weight gradients are sums over atoms in another
    order and agree to rounding."""
    import torch
    from nmrgnn_amd import synth
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    b = synth.make_batch(graphs, atoms, 16, 10, 0.08, seed=F + graphs)
    N = graphs * atoms
    rng = np.random.default_rng(7)
    new_of_old = np.concatenate([g * atoms + rng.permutation(atoms) for g in range(graphs)])      # old row -> new row
    old_of_new = np.empty(N, np.int64)
    old_of_new[new_of_old] = np.arange(N)
    nl_new = new_of_old[b["nlist"][old_of_new]]
    # padded slots point at LOCAL atom 0 of their graph in the reference's convention (index 0 + offset): keep that form
    pad = b["edges"][old_of_new] == 0
    nl_new[pad] = (np.arange(N)[:, None] // atoms * atoms + 0 * nl_new)[pad]
    bp = dict(atoms=b["atoms"][old_of_new], nlist=nl_new.astype(np.int32), edges=b["edges"][old_of_new],
              inv_degree=b["inv_degree"][old_of_new])
    eng = Engine(make_hp(atom_feature_size=F), 10, device=gpu_device, seed=5)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    gp = GraphBatch(bp["atoms"], bp["nlist"], bp["edges"], bp["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    peaks, peaks_p = eng.forward(gb).cpu().numpy(), eng.forward(gp).cpu().numpy()
    np.testing.assert_array_equal(peaks_p[new_of_old], peaks)
    # training pass with the draws handed in (noise per slot, dropout per atom), permuted alike
    xi = eng.randn(N * 16, seed=3).reshape(N, 16)
    Fh = F // 2
    mask = eng.dropout_mask(N * Fh, seed=4).reshape(N, Fh)
    dpe = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(gpu_device)
    idx = torch.from_numpy(old_of_new).to(gpu_device)
    pk = eng.forward(gb, training=True, noise=xi.reshape(-1), dropout_mask=mask.reshape(-1)).clone()
    eng.backward(dpe)
    g0 = eng.params.grads_dict()
    pkp = eng.forward(gp, training=True, noise=xi[idx].reshape(-1).contiguous(), dropout_mask=mask[idx].reshape(-1).contiguous()).clone()
    eng.backward(dpe[idx].contiguous())
    g1 = eng.params.grads_dict()
    np.testing.assert_array_equal(pkp.cpu().numpy()[new_of_old], pk.cpu().numpy())
    for k in g0:
        scale = max(np.abs(g0[k]).max(), 1e-30)
        assert np.abs(g1[k] - g0[k]).max() <= 2e-5 * scale, k


@pytest.mark.parametrize("F", [64, 256])
@pytest.mark.parametrize("k", [-9, 6])
def test_backward_is_homogeneous_in_powers_of_two(gpu_device, F, k):
    """The backward is linear in the upstream gradient, and every scale the fp16-piece kernels choose for their gradient operands
    is a power of two taken from a maximum (edge backward: per call; FC block and window kernels: per row; generic GEMMs: per
    call), so multiplying dpeaks by 2^k must multiply every weight gradient by exactly 2^k — the pieces, products and sums are
    the same bits with another exponent."""
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    b = small_batch(6, 150, seed=9)
    eng = Engine(make_hp(atom_feature_size=F), 10, device=gpu_device, seed=2)
    randomize_biases(eng)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    N, K = b["edges"].shape
    xi = eng.randn(N * K, seed=5)
    mask = eng.dropout_mask(N * (F // 2), seed=6)
    dpe = torch.from_numpy(np.random.default_rng(1).standard_normal(N).astype(np.float32)).to(gpu_device)
    out = []
    for scale in (1.0, 2.0 ** k):
        eng.forward(gb, training=True, noise=xi, dropout_mask=mask)
        eng.backward(dpe * scale)
        out.append(eng.params.grads_dict())
    for name in out[0]:
        np.testing.assert_array_equal(out[1][name], out[0][name] * np.float32(2.0 ** k), err_msg=name)


def test_padded_slot_index_is_irrelevant(gpu_device):
    """KAT-3: changing nlist in a padded (edges == 0) slot must not change any peak."""
    from nmrgnn_amd.graph import GraphBatch
    hp, b, eng, sd, gb, std, avg = _setup(gpu_device, CONFIGS[0], n_graphs=1, n_atoms=64)
    base = eng.forward(gb).cpu().numpy()
    nl = b["nlist"].copy()
    pad = b["edges"] == 0
    assert pad.any()
    nl[pad] = 17
    gb2 = GraphBatch(b["atoms"], nl, b["edges"], b["inv_degree"], device=gpu_device)
    np.testing.assert_array_equal(eng.forward(gb2).cpu().numpy(), base)


def test_loss_and_adam_match_oracle(gpu_device):
    import torch
    from oracle import nmrgnn_oracle as O
    hp, b, eng, sd, gb, std, avg = _setup(gpu_device, CONFIGS[1])
    N = gb.N
    rng = np.random.default_rng(2)
    pred = rng.standard_normal(N).astype(np.float32)
    w = (rng.random(N) > 0.3).astype(np.float32)
    w[: gb.graph_ptr_host[1]] = 0.0          # a graph with zero total weight -> divide_no_nan
    y = b["y"]
    dev = gpu_device
    loss, dpred = eng.loss_l2(gb, torch.from_numpy(y).to(dev), torch.from_numpy(w).to(dev),
                              torch.from_numpy(pred).to(dev))
    rl, rg = O.batch_loss_s1(y, w, pred, gb.graph_ptr_host)
    assert abs(float(loss.cpu()) - rl) < 1e-5 * max(1.0, abs(rl))
    np.testing.assert_allclose(dpred.cpu().numpy(), rg, rtol=1e-5, atol=1e-7)
    # Adam: three steps on a fixed gradient
    p0 = eng.params.flat.cpu().numpy().astype(np.float64)
    g = rng.standard_normal(p0.shape).astype(np.float32)
    eng.params.grad.copy_(torch.from_numpy(g).to(dev))
    m = np.zeros_like(p0); v = np.zeros_like(p0); p = p0.copy()
    for t in range(1, 4):
        eng.adam_step(lr=1e-3)
        p, m, v = O.adam_step(p, g.astype(np.float64), m, v, t, lr=1e-3)
    np.testing.assert_allclose(eng.params.flat.cpu().numpy(), p, rtol=2e-5, atol=1e-7)


def test_golden_fixture_bench_arch(gpu_device):
    """committed golden vectors (tests/golden/golden_f64.npz, generated by tests/golden/make_golden.py)"""
    import os
    import torch
    from oracle import nmrgnn_oracle as O
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_f64.npz"))
    hp = make_hp(atom_feature_size=64, edge_feature_size=3, edge_hidden_size=128)
    p = O.init_params(hp_to_oracle(hp), 10, seed=int(g["param_seed"]), dtype=np.float32,
                      bias_scale=float(g["bias_scale"]))
    eng = Engine(hp, 10, device=gpu_device)
    eng.params.load_state_dict(p)
    gb = GraphBatch(g["atoms"], g["nlist"], g["edges"], g["inv_degree"], graph_ptr=g["graph_ptr"],
                    device=gpu_device)
    peaks = eng.forward(gb).cpu().numpy()
    assert np.max(np.abs(peaks - g["peaks"])) < PEAK_ATOL
    dev = gpu_device
    pt = eng.forward(gb, training=True, noise=torch.from_numpy(g["noise"]).to(dev),
                     dropout_mask=torch.from_numpy(g["dropout_mask"] / 0.8).to(dev))
    assert np.max(np.abs(pt.cpu().numpy() - g["peaks_train"])) < PEAK_ATOL
    eng.backward(torch.from_numpy(g["dpeaks"]).to(dev))
    grads = eng.params.grads_dict()
    for k in grads:
        if "g:" + k in g.files:
            assert rel_err(grads[k], g["g:" + k]) < GRAD_RTOL, k
        s = g["gs:" + k]
        assert abs(grads[k].sum() - s[0]) <= GRAD_RTOL * s[1] + 1e-6, k
        assert abs(np.abs(grads[k]).max() - s[2]) <= GRAD_RTOL * s[2] + 1e-7, k


def test_ring_fixture_reference_test_graph(gpu_device):
    """the reference's own test graph (reference tests/test_nmrgnn.py:197-223): 5 atoms, ring, 16 elements"""
    import os
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ring.npz"))
    hp = make_hp(atom_feature_size=32, edge_feature_size=2, edge_hidden_size=16, mp_layers=2,
                 fc_layers=2, edge_fc_layers=2)
    eng = Engine(hp, 16, g["peak_std"], g["peak_avg"], device=gpu_device)
    eng.params.load_state_dict({k[2:]: g[k] for k in g.files if k.startswith("p:")})
    gb = GraphBatch(g["atoms"], g["nlist"], g["edges"], g["inv_degree"], device=gpu_device)
    peaks = eng.forward(gb)
    assert peaks.shape == (5,)
    assert np.max(np.abs(peaks.cpu().numpy() - g["peaks"])) < PEAK_ATOL
    hp0 = make_hp(atom_feature_size=32, edge_feature_size=2, edge_hidden_size=16, mp_layers=2,
                  fc_layers=2, edge_fc_layers=2, noise=0.0, dropout=False)
    eng0 = Engine(hp0, 16, g["peak_std"], g["peak_avg"], device=gpu_device)
    eng0.params.load_state_dict({k[2:]: g[k] for k in g.files if k.startswith("p:")})
    eng0.forward(gb, training=True)
    eng0.backward(torch.ones(5, device=gpu_device))
    grads = eng0.params.grads_dict()
    for k in grads:
        assert rel_err(grads[k], g["g:" + k]) < GRAD_RTOL, k


def test_full_size_fused_equals_layered_and_batch_invariance(gpu_device, monkeypatch):
    """BASELINE configs[1] size (512 x 256 atoms): size-independent properties —
    (i) the fused persistent edge kernels and the one-launch-per-layer path agree;
    (ii) any sub-batch of graphs gives the same peaks as inside the full batch;
    (iii) the oracle agrees on a sample of graphs."""
    import torch
    from oracle import nmrgnn_oracle as O
    from nmrgnn_amd import synth
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    hp = make_hp(atom_feature_size=64, edge_feature_size=3, edge_hidden_size=128)
    b = synth.make_batch(512, 256, 16, 10, 0.05, seed=42)
    eng = Engine(hp, 10, device=gpu_device, seed=1234)
    sd = randomize_biases(eng, scale=0.05)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"],
                    device=gpu_device)
    N, K = b["edges"].shape
    xi = eng.randn(N * K, seed=7)
    mask = eng.dropout_mask(N * 32, seed=8)
    dpe = torch.from_numpy(np.random.default_rng(0).standard_normal(N).astype(np.float32)).to(gpu_device)

    def run():
        pk = eng.forward(gb, training=True, noise=xi, dropout_mask=mask).clone()
        eng.backward(dpe)
        return pk.cpu().numpy(), eng.params.grads_dict()

    pk_f, g_f = run()
    monkeypatch.setenv("NG_EDGE_PATH", "layered")
    pk_l, g_l = run()
    monkeypatch.delenv("NG_EDGE_PATH")
    assert np.max(np.abs(pk_f - pk_l)) < 5e-5
    for k in g_f:
        assert rel_err(g_f[k], g_l[k]) < GRAD_RTOL, k
    # sub-batch invariance + oracle on 2 graphs out of 512
    inf = eng.forward(gb).cpu().numpy()
    for gidx in (0, 311):
        sl = slice(gidx * 256, (gidx + 1) * 256)
        sub = (b["atoms"][sl], b["nlist"][sl] - gidx * 256, b["edges"][sl], b["inv_degree"][sl])
        part = eng.forward(GraphBatch(*sub, device=gpu_device)).cpu().numpy()
        np.testing.assert_allclose(part, inf[sl], rtol=0, atol=1e-5)
        ref = O.gnn_forward(sub, sd, hp_to_oracle(hp))
        assert np.max(np.abs(part - ref)) < PEAK_ATOL


@pytest.mark.parametrize("edge_table", [False, True])
def test_full_size_gradients_match_oracle(gpu_device, edge_table):
    """(edge_table=True: the opt-in edge function table of csrc/edge_table.hip — the same bounds, at the full size)
    BASELINE configs[2] at its full size (512 x 256 atoms, F=64): peaks AND every gradient tensor of
    the fused training step against the float64 oracle run over ALL 512 graphs (graphs are independent,
    so the batch gradient is the sum of per-graph oracle gradients; tests/helpers.py runs them in worker
    processes), with the GPU's own noise / dropout draws fed to the oracle."""
    import torch
    from nmrgnn_amd import synth
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    from helpers import oracle_batch_forward_backward
    hp = make_hp(atom_feature_size=64, edge_feature_size=3, edge_hidden_size=128)
    b = synth.make_batch(512, 256, 16, 10, 0.05, seed=42)
    eng = Engine(hp, 10, device=gpu_device, seed=1234)
    eng.edge_table = edge_table
    sd = randomize_biases(eng, scale=0.05)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"],
                    device=gpu_device)
    N, K = b["edges"].shape
    xi = eng.randn(N * K, seed=7)
    mask = eng.dropout_mask(N * 32, seed=8)
    # the loss gradient of the bench step: NameLoss s=1 on the batch's labels (1/G per graph)
    peaks = eng.forward(gb, training=True, noise=xi, dropout_mask=mask)
    assert (eng.tape.table is not None) == edge_table
    loss, dpe = eng.loss_l2(gb, torch.from_numpy(b["y"]).to(gpu_device), torch.from_numpy(b["w"]).to(gpu_device),
                            peaks)
    eng.backward(dpe)
    pk = peaks.cpu().numpy()
    grads = eng.params.grads_dict()
    ref_pk, ref_g = oracle_batch_forward_backward(
        b, sd, hp_to_oracle(hp), dpe.cpu().numpy().astype(np.float64),
        xi=xi.cpu().numpy().reshape(N, K).astype(np.float64),
        mask=(mask.cpu().numpy().reshape(N, 32) > 0).astype(np.float64), workers=16)
    assert np.max(np.abs(pk - ref_pk)) < PEAK_ATOL
    errs = {k: rel_err(grads[k], g) for k, g in ref_g.items()}
    print("full-size gradient rel. errors:", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < GRAD_RTOL, errs


def test_empty_and_tiny_graphs(gpu_device):
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    from oracle import nmrgnn_oracle as O
    hp = make_hp(atom_feature_size=64, edge_feature_size=3, edge_hidden_size=128)
    eng = Engine(hp, 10, device=gpu_device)
    sd = eng.params.state_dict()
    # one atom, no real neighbour: every slot padded, inv_degree 0
    atoms = np.zeros((1, 10), np.float32); atoms[0, 4] = 1
    tup = (atoms, np.zeros((1, 16), np.int32), np.zeros((1, 16), np.float32), np.zeros(1, np.float32))
    pk = eng.forward(GraphBatch(*tup, device=gpu_device)).cpu().numpy()
    np.testing.assert_allclose(pk, O.gnn_forward(tup, sd, hp_to_oracle(hp)), atol=PEAK_ATOL)
    empty = (np.zeros((0, 10), np.float32), np.zeros((0, 16), np.int32), np.zeros((0, 16), np.float32),
             np.zeros(0, np.float32))
    assert eng.forward(GraphBatch(*empty, device=gpu_device)).shape == (0,)


@pytest.mark.parametrize("s", [0.0, 0.3, 1.0])
def test_name_loss_balance_vs_oracle(gpu_device, s):
    """ng_loss_name: s*l2 + (1-s)*(1-r) per graph (nmrgnn/losses.py:30-39) incl. an empty graph and a
    graph with zero total weight; carbon-like shifts (120 +- 3) stress the moment cancellation."""
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    from oracle import nmrgnn_oracle as O
    hp = make_hp()
    b = small_batch(4, 33, seed=2)
    ptr = [0, 40, 40, 97, 132]
    N = ptr[-1]
    eng = Engine(hp, 10, device=gpu_device, seed=1)
    gb = GraphBatch(b["atoms"][:N], b["nlist"][:N] % N, b["edges"][:N], b["inv_degree"][:N], graph_ptr=ptr,
                    device=gpu_device)
    rng = np.random.default_rng(5)
    y = (rng.standard_normal(N) * 3 + 120).astype(np.float32)
    pred = (y + rng.standard_normal(N)).astype(np.float32)
    w = ((rng.random(N) > 0.3) * rng.random(N)).astype(np.float32)
    w[97:] = 0.0
    ty, tw, tp = (torch.from_numpy(x).to(gpu_device) for x in (y, w, pred))
    loss, dpred = eng.loss_name(gb, ty, tw, tp, s)
    ref_l, ref_g = O.batch_loss_name(y, w, pred, ptr, s)
    assert loss.item() == pytest.approx(ref_l, rel=2e-6, abs=1e-7)
    assert rel_err(dpred.cpu().numpy(), ref_g) < 1e-5
    if s == 1.0:
        l2, d2 = eng.loss_l2(gb, ty, tw, tp)
        assert l2.item() == pytest.approx(loss.item(), rel=1e-6)
        assert rel_err(d2.cpu().numpy(), dpred.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("F", [64, 256])
def test_gradients_with_upstream_gradients_spanning_six_decades(gpu_device, monkeypatch, F):
    """A few labelled atoms with O(1) gradients, the rest 1e-6 .. 1e-1 of that or exactly zero (NMR batches mix labelled
    and unlabelled atoms).  The fp16-piece kernels scale gradient operands per row / per tile by powers of two
    (mp_win_bwd.hip: the h operand of the dw product takes the inverse of the B rows' scales); the round-2 advisor asked
    whether small rows then lose their contribution.  Every gradient tensor against the float64 oracle: within 5e-5 of the
    tensor's largest entry, and no worse than the f32-input MFMA kernels on the same inputs times 8 (+ a rounding floor): two
    fp16 pieces carry 22 mantissa bits, so a piece product is good to 2^-21 where the f32-input one is good to 2^-24, and
    both paths see the same cancellation in these sums.  (The factor was 4 while the fp32 path's own error on edge_fc/3
    at F = 256 sat at 5e-6; with the shorter softplus of round 4 that error fell to 2.7e-6 and the piece path's
    1.7e-5 - 2.1e-5 before - no longer fitted 4 x.)"""
    import torch
    from oracle import nmrgnn_oracle as O
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    hp = make_hp(atom_feature_size=F, edge_feature_size=3, edge_hidden_size=128)
    b = small_batch(6, 120, seed=3)
    rng = np.random.default_rng(5)
    std = rng.uniform(0.5, 2.0, 10).astype(np.float32)
    avg = rng.uniform(-1.0, 1.0, 10).astype(np.float32)
    N, K = b["edges"].shape
    dpeaks = (rng.standard_normal(N) * 10.0 ** rng.uniform(-6, 0, N)).astype(np.float32)
    dpeaks[rng.random(N) < 0.5] = 0.0
    res = {}
    for mode in ("f16x2", "fp32"):
        monkeypatch.setenv("NG_GEMM_MATH", mode)
        monkeypatch.setenv("NG_EDGE_MATH", mode)
        eng = Engine(hp, 10, std, avg, device=gpu_device, seed=11)
        sd = randomize_biases(eng)
        gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
        eng.forward(gb, training=True, noise=torch.zeros(N * K, device=gpu_device),
                    dropout_mask=torch.full((N * F // 2,), 1.25, device=gpu_device))
        eng.backward(torch.from_numpy(dpeaks).to(gpu_device))
        res[mode] = {k: v.copy() for k, v in eng.params.grads_dict().items()}
    _, ref = O.gnn_forward_backward((b["atoms"], b["nlist"], b["edges"], b["inv_degree"]), sd, hp_to_oracle(hp), dpeaks, std, avg,
                                    training=True, noise=np.zeros((N, K)), dropout_mask=np.ones((N, F // 2)))
    bad = {}
    for k, g in ref.items():
        g = np.asarray(g, dtype=np.float64)
        mx = np.abs(g).max()
        e2 = np.abs(res["f16x2"][k] - g).max() / mx
        e1 = np.abs(res["fp32"][k] - g).max() / mx
        if e2 > 5e-5 or e2 > 8.0 * e1 + 2e-6:
            bad[k] = (e2, e1)
    assert not bad, bad
