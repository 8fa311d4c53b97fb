"""Head forward + L2 NameLoss + head backward as one launch (csrc/head_ops.hip: head_loss_kernel; ng_head_loss_bwd, ABI 9;
reference nmrgnn/model.py:266-273, nmrgnn/losses.py:30-39): against the three-call chain it replaces.  Peaks and every
gradient downstream of dg carry the chain's bits; dWout / dbout and the mean over graphs are summed in another order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hp(F=64):
    from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
    return declare_gnn_space(HyperParameters(atom_feature_size=F, edge_feature_size=3, edge_hidden_size=128, mp_layers=4,
                                             fc_layers=4, edge_fc_layers=4))


def _ragged(sizes, K=16, seed=11):
    """graphs of different sizes in one concatenated tuple"""
    from nmrgnn_amd import synth
    parts, off = [], 0
    ptr = [0]
    for k, n in enumerate(sizes):
        b = synth.make_batch(1, n, K, 10, 0.08, seed=seed + k)
        parts.append((b["atoms"], b["nlist"] + off, b["edges"], b["inv_degree"], b["y"]))
        off += n
        ptr.append(off)
    cat = lambda i: np.concatenate([p[i] for p in parts])
    rng = np.random.default_rng(seed)
    w = (rng.uniform(size=off) < 0.8).astype(np.float32) * rng.uniform(0.5, 2.0, size=off).astype(np.float32)
    return (cat(0), cat(1).astype(np.int32), cat(2), cat(3)), np.asarray(ptr, np.int32), cat(4), w


def _pair(dev, F=64, dropout=True):
    from nmrgnn_amd.engine import Engine
    from tests.helpers import randomize_biases
    ea, eb = Engine(_hp(F), 10, device=dev, seed=31), Engine(_hp(F), 10, device=dev, seed=31)
    for e in (ea, eb):
        randomize_biases(e)
        e.use_dropout = dropout
        e.peak_std.copy_(torch.linspace(0.5, 40.0, 10, device=dev))
        e.peak_avg.copy_(torch.linspace(-3.0, 120.0, 10, device=dev))
    ea.fuse_head_loss, eb.fuse_head_loss = True, False
    return ea, eb


def _check(ea, eb, raw, gp, y, w, gw, dev, expect_fused=True):
    from nmrgnn_amd.graph import GraphBatch
    t = lambda a: torch.as_tensor(np.asarray(a)).to(device=dev, dtype=torch.float32)
    ba, bb = GraphBatch(*raw, graph_ptr=gp, device=dev), GraphBatch(*raw, graph_ptr=gp, device=dev)
    pa = ea.forward(ba, training=True, seed=9, loss=(t(y), t(w), gw))
    pb = eb.forward(bb, training=True, seed=9, loss=(t(y), t(w), gw))
    assert (ea.tape.dg is not None) == expect_fused and eb.tape.dg is None
    la, lb = ea.tape.loss, eb.tape.loss          # the fused launch's loss is complete once backward() has run
    ea.backward(None)
    eb.backward(None)
    torch.cuda.synchronize()
    assert torch.equal(pa, pb)
    assert abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb)) + 1e-30, (la, lb)
    ga, gb = ea.params.grads_dict(), eb.params.grads_dict()
    for k in ga:
        if k.startswith("out/") and expect_fused:
            scale = float(np.abs(gb[k]).max()) + 1e-30
            assert float(np.abs(ga[k] - gb[k]).max()) <= 2e-6 * scale, k
        else:
            assert np.array_equal(ga[k], gb[k]), k
    return la


@pytest.mark.parametrize("F", [64, 256])
@pytest.mark.parametrize("dropout", [True, False])
def test_fused_head_loss_is_the_three_call_chain(gpu_device, F, dropout):
    ea, eb = _pair(gpu_device, F, dropout)
    raw, gp, y, w = _ragged([256, 40, 131, 256, 77, 200, 9])
    loss = _check(ea, eb, raw, gp, y, w, 1.0, gpu_device)
    assert np.isfinite(float(loss)) and float(loss) > 0


def test_fused_head_loss_many_graphs_and_shard_weight(gpu_device):
    # more graphs than workgroups (several graphs per workgroup, the last one short) and a data-parallel shard weight
    ea, eb = _pair(gpu_device)
    raw, gp, y, w = _ragged([24 + (7 * k) % 40 for k in range(700)], seed=5)
    _check(ea, eb, raw, gp, y, w, 0.375, gpu_device)


def test_unlabelled_graph_and_long_graph(gpu_device):
    ea, eb = _pair(gpu_device)
    raw, gp, y, w = _ragged([64, 64, 64])
    w[64:128] = 0.0                                   # a graph without a labelled atom: loss 0, gradient 0 (sum w == 0)
    _check(ea, eb, raw, gp, y, w, 1.0, gpu_device)
    # a graph beyond the rows a workgroup can own, and more graphs than one round of workgroups: the engine takes the three calls
    raw, gp, y, w = _ragged([300, 50])
    _check(ea, eb, raw, gp, y, w, 1.0, gpu_device, expect_fused=False)
    raw, gp, y, w = _ragged([200] * 600, seed=3)
    _check(ea, eb, raw, gp, y, w, 1.0, gpu_device, expect_fused=False)


def test_backward_with_an_explicit_gradient_is_refused_after_the_fused_forward(gpu_device):
    from nmrgnn_amd.graph import GraphBatch
    ea, _ = _pair(gpu_device)
    raw, gp, y, w = _ragged([50, 60])
    t = lambda a: torch.as_tensor(np.asarray(a)).to(device=gpu_device, dtype=torch.float32)
    ea.forward(GraphBatch(*raw, graph_ptr=gp, device=gpu_device), training=True, seed=1, loss=(t(y), t(w), 1.0))
    with pytest.raises(RuntimeError):
        ea.backward(torch.ones(110, device=gpu_device))


def test_trainer_trajectory_fused_against_three_calls(gpu_device):
    from nmrgnn_amd.graph import GraphBatch
    from nmrgnn_amd.train import Trainer
    ea, eb = _pair(gpu_device)
    ta, tb = Trainer(ea, lr=1e-3), Trainer(eb, lr=1e-3)
    raw, gp, y, w = _ragged([256, 256, 100])
    t = lambda a: torch.as_tensor(np.asarray(a)).to(device=gpu_device, dtype=torch.float32)
    for step in range(4):
        la = ta.step(GraphBatch(*raw, graph_ptr=gp, device=gpu_device), t(y), t(w))
        lb = tb.step(GraphBatch(*raw, graph_ptr=gp, device=gpu_device), t(y), t(w))
        torch.cuda.synchronize()
        assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb)), step
    d = float((ea.params.flat - eb.params.flat).abs().max())
    assert d <= 2e-5, d                                   # Adam steps of 1e-3: the out-layer sums differ in their last bits


def test_small_call_embedding_gradient_rides_with_the_deferred_reductions(gpu_device):
    """molecule-sized backward: dWemb is a job of the batched second stage (reduce.cuh: outer job), no launch of its own —
    against the two-stage form of an engine that does not defer"""
    from nmrgnn_amd.graph import GraphBatch
    ea, eb = _pair(gpu_device)
    ea.fuse_head_loss = eb.fuse_head_loss = False
    eb.defer_reductions = False
    raw, gp, y, w = _ragged([200, 56])
    t = lambda a: torch.as_tensor(np.asarray(a)).to(device=gpu_device, dtype=torch.float32)
    for e in (ea, eb):
        e.forward(GraphBatch(*raw, graph_ptr=gp, device=gpu_device), training=True, seed=4, loss=(t(y), t(w), 1.0))
        e.backward(None)
    torch.cuda.synchronize()
    ga, gb = ea.params.grads_dict(), eb.params.grads_dict()
    for k in ga:
        scale = float(np.abs(gb[k]).max()) + 1e-30
        if k == "embed/kernel":
            assert float(np.abs(ga[k] - gb[k]).max()) <= 2e-6 * scale
            assert float(np.abs(gb[k]).max()) > 0
        else:
            assert np.array_equal(ga[k], gb[k]), k


def test_understated_graph_length_poisons_the_loss_and_stays_in_bounds(gpu_device):
    """C-ABI caller error: max_graph_atoms smaller than graph_ptr's longest graph.  The launch must stay inside its LDS arrays
    (rows beyond the 256 a workgroup can hold are left unwritten) and the loss comes back NaN."""
    import ctypes as C
    from nmrgnn_amd import _lib
    from nmrgnn_amd._lib import ptr
    dev = gpu_device
    ctx = _lib.get_context(dev.index)
    N, Fh, Cn = 400, 32, 10
    g = torch.randn(N, Fh, device=dev)
    atoms = torch.zeros(N, Cn, device=dev); atoms[:, 3] = 1
    W, b = torch.randn(Fh, Cn, device=dev), torch.randn(Cn, device=dev)
    std, avg = torch.ones(Cn, device=dev), torch.zeros(Cn, device=dev)
    gptr = torch.tensor([0, 300, 400], dtype=torch.int32, device=dev)
    y, w = torch.randn(N, device=dev), torch.ones(N, device=dev)
    nb = ctx.lib.ng_head_loss_blocks(ctx.handle, 2, Fh, Cn, 200)          # the caller claims 200 atoms per graph
    assert nb == 2
    peaks, dg = torch.zeros(N, device=dev), torch.zeros(N, Fh, device=dev)
    part = torch.zeros(nb, Fh * Cn + Cn + 1, device=dev)
    dW, db, loss = torch.zeros(Fh, Cn, device=dev), torch.zeros(Cn, device=dev), torch.zeros(1, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ctx.check(ctx.lib.ng_head_loss_bwd(ctx.handle, st, N, 2, Fh, Cn, 200, ptr(g), 1, 0, 1.0, None, ptr(W), ptr(b), ptr(atoms), ptr(std),
                                       ptr(avg), ptr(gptr), ptr(y), ptr(w), 1.0, ptr(peaks), ptr(dg), ptr(part)), "head_loss_bwd")
    ctx.check(ctx.lib.ng_head_loss_reduce(ctx.handle, st, ptr(part), nb, Fh, Cn, ptr(dW), ptr(db), ptr(loss)), "head_loss_reduce")
    torch.cuda.synchronize()
    assert torch.isnan(loss).all()
    assert torch.isfinite(peaks).all() and torch.isfinite(dg).all()
    assert float(peaks[256:300].abs().max()) == 0.0          # rows a workgroup cannot hold: untouched
