"""Run-to-run determinism: the engine uses no floating-point atomics and fixed-order reductions, so the same inputs
must give bit-identical outputs and gradients every time.  (Round 2 found the training variant of the split-operand
edge forward returning slightly different e from run to run: inline-asm bf16 conversions that the scheduler had moved
into an MFMA chain without the hazard wait states — csrc/h2_common.cuh.  This test keeps that class of bug out.)"""
import numpy as np
import pytest

from helpers import make_hp, randomize_biases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_edges", [3159, 100000, 1 << 20])
@pytest.mark.parametrize("train", [False, True])
def test_edge_forward_is_bitwise_reproducible(gpu_device, n_edges, train):
    import torch
    from nmrgnn_amd._lib import ptr, ptr_array
    from nmrgnn_amd.engine import Engine
    eng = Engine(make_hp(atom_feature_size=64), 10, device=gpu_device, seed=2)
    randomize_biases(eng)
    P = eng.params
    W = [P[f"edge_fc/{t}/kernel"] for t in range(4)]
    B = [P[f"edge_fc/{t}/bias"] for t in range(4)]
    d = torch.from_numpy(np.random.default_rng(0).uniform(0.05, 0.5, n_edges).astype(np.float32)).to(gpu_device)

    def run():
        e = torch.empty(n_edges, 3, device=gpu_device)
        z = torch.empty(3, n_edges, 128, device=gpu_device) if train else None
        eng._ck(eng.lib.ng_edge_mlp_fwd(eng.ctx.handle, eng._st(), n_edges, 128, 3, 4, 1, ptr(d), ptr(d), ptr(eng.centers),
                                        eng.gap, ptr_array(W), ptr_array(B), ptr(e), ptr(z)), "edge fwd")
        return e, z

    e0, z0 = run()
    for _ in range(10):
        e1, z1 = run()
        assert torch.equal(e0, e1)
        if train:
            assert torch.equal(z0, z1)


@pytest.mark.parametrize("F", [64, 256])
def test_training_step_is_bitwise_reproducible(gpu_device, F):
    """forward(training) + backward on 64 graphs x 256 atoms, five times: identical peaks and gradients"""
    import torch
    from nmrgnn_amd import synth
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    hp = make_hp(atom_feature_size=F)
    b = synth.make_batch(64, 256, 16, 10, 0.05, seed=3)
    eng = Engine(hp, 10, device=gpu_device, seed=9)
    randomize_biases(eng)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    N, K = b["edges"].shape
    xi = eng.randn(N * K, seed=5)
    mask = eng.dropout_mask(N * (F // 2), seed=6)
    dpe = torch.from_numpy(np.random.default_rng(1).standard_normal(N).astype(np.float32)).to(gpu_device)
    ref = None
    for _ in range(5):
        pk = eng.forward(gb, training=True, noise=xi, dropout_mask=mask).clone()
        eng.backward(dpe)
        cur = (pk, eng.params.grad.clone())
        if ref is None:
            ref = cur
        else:
            assert torch.equal(cur[0], ref[0])
            assert torch.equal(cur[1], ref[1])


def test_frozen_weight_cache_tracks_weight_changes(gpu_device):
    """ng_weights_frozen keeps the packed weight images across calls; every sanctioned way of changing weights
    (load_state_dict, an Adam step) must invalidate them, unfreezing must switch the cache off — the outputs always equal
    those of an engine that re-packs on every call."""
    import torch
    from nmrgnn_amd import synth
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    for F in (64, 256):
        hp = make_hp(atom_feature_size=F)
        b = synth.make_batch(3, 50, 16, 10, 0.1, seed=3)
        gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
        ref = Engine(hp, 10, device=gpu_device, seed=9)          # never frozen
        eng = Engine(hp, 10, device=gpu_device, seed=9)
        eng.freeze_weights(True)
        assert torch.equal(eng.forward(gb), ref.forward(gb))
        assert torch.equal(eng.forward(gb), ref.forward(gb))     # second call: served from the cache
        sd = randomize_biases(ref, seed=4)
        eng.params.load_state_dict(sd)                           # announces the change itself
        assert torch.equal(eng.forward(gb), ref.forward(gb))
        dpe = torch.ones(gb.N, device=gpu_device)
        for e in (eng, ref):
            e.forward(gb, training=True, seed=3)
            e.backward(dpe)
            e.adam_step(lr=1e-2)                                 # ng_adam_step bumps the weight version
        assert torch.equal(eng.forward(gb), ref.forward(gb))
        eng.params["fc/0/bias"].add_(0.5)                        # a raw write into a view needs weights_changed()
        ref.params["fc/0/bias"].add_(0.5)
        eng.weights_changed()
        assert torch.equal(eng.forward(gb), ref.forward(gb))
        eng.freeze_weights(False)
        assert torch.equal(eng.forward(gb), ref.forward(gb))
        # the cache is keyed by weight addresses and shared by everything on the device: two frozen models must not see
        # each other's images, and a model built where a deleted frozen one lived must not inherit them
        eng.freeze_weights(True)
        other, other_ref = Engine(hp, 10, device=gpu_device, seed=21), Engine(hp, 10, device=gpu_device, seed=21)
        other.freeze_weights(True)
        for _ in range(2):
            assert torch.equal(eng.forward(gb), ref.forward(gb))
            assert torch.equal(other.forward(gb), other_ref.forward(gb))
        del eng, other
        torch.cuda.empty_cache()
        for seed in (33, 34):
            late, late_ref = Engine(hp, 10, device=gpu_device, seed=seed), Engine(hp, 10, device=gpu_device, seed=seed)
            late.freeze_weights(True)
            assert torch.equal(late.forward(gb), late_ref.forward(gb))
            assert torch.equal(late.forward(gb), late_ref.forward(gb))
            del late, late_ref


def test_backward_inside_a_frozen_window_uses_each_layers_own_weights(gpu_device):
    """Round-2 advisor finding: the frozen-weight cache was keyed by the raw W pointer, and the MPLayer backward hands the
    GEMMs a weight matrix it has just repacked into the context's scratch — the same address for every layer.  With
    ng_weights_frozen on for a whole forward + backward (the C ABI does not forbid it) layer l-1 multiplied with layer
    l's cached image.  Scratch sources are never cache keys now: gradients inside a frozen window equal the unfrozen ones
    bit for bit, at the default width (layered GEMM path) and at F = 64."""
    import torch
    from nmrgnn_amd import synth
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    for F in (256, 64):
        hp = make_hp(atom_feature_size=F)
        b = synth.make_batch(40, 128, 16, 10, 0.1, seed=6)        # 5120 atoms: the split-operand GEMMs take the products
        gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
        ref = Engine(hp, 10, device=gpu_device, seed=9)
        eng = Engine(hp, 10, device=gpu_device, seed=9)
        dpe = torch.randn(gb.N, device=gpu_device, generator=torch.Generator(device=gpu_device).manual_seed(1))
        ref.forward(gb, training=True, seed=5)
        ref.backward(dpe)
        lib, h = eng.lib, eng.ctx.handle
        for rep in range(2):                                       # second pass: everything cacheable IS cached
            assert lib.ng_weights_frozen(h, eng._id) == 0
            try:
                eng._forward(gb, True, None, None, 5)
                eng.backward(dpe)
            finally:
                lib.ng_weights_frozen(h, 0)
            assert torch.equal(eng.params.grad, ref.params.grad), (F, rep)


@pytest.mark.parametrize("F", [64, 256])
def test_deferred_weight_gradient_sums_equal_the_eager_ones(gpu_device, F):
    """Engine.backward queues the second-stage sums of the node-side weight gradients and runs them in one launch
    (ng_defer_reductions / ng_flush_reductions, ABI 5): every gradient must carry the bits of the eager form, also when
    the queue is longer than one batch and on the second call (arena reuse)."""
    import torch
    from nmrgnn_amd import synth
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    hp = make_hp(atom_feature_size=F)
    b = synth.make_batch(48, 200, 16, 10, 0.05, seed=4)
    eng = Engine(hp, 10, device=gpu_device, seed=11)
    randomize_biases(eng)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=gpu_device)
    N, K = b["edges"].shape
    xi = eng.randn(N * K, seed=5)
    mask = eng.dropout_mask(N * (F // 2), seed=6)
    dpe = torch.from_numpy(np.random.default_rng(2).standard_normal(N).astype(np.float32)).to(gpu_device)
    grads = {}
    for mode in (False, True, True):
        eng.defer_reductions = mode
        eng.params.grad.fill_(12345.0)       # a gradient the backward did not (re)write would keep the sentinel
        eng.forward(gb, training=True, noise=xi, dropout_mask=mask)
        eng.backward(dpe)
        g = eng.params.grad.clone()
        # (the flat buffer pads every tensor to 16 bytes: only the views are gradients)
        for name, gv in eng.params.grad_views.items():
            assert torch.isfinite(gv).all() and not bool((gv == 12345.0).any()), name
        if mode in grads:
            assert torch.equal(grads[mode], g)
        grads[mode] = g
    assert torch.equal(grads[False], grads[True])
    # C ABI: switching deferral off flushes what is queued
    lib, h, st = eng.lib, eng.ctx.handle, eng._st()
    eng._ck(lib.ng_defer_reductions(h, st, 1), "on")
    eng._ck(lib.ng_flush_reductions(h, st), "empty flush")
    eng._ck(lib.ng_defer_reductions(h, st, 0), "off")


def test_fused_draws_carry_the_bits_of_the_two_launch_forms(gpu_device):
    """ng_add_noise == ng_randn + ng_add_scaled and ng_head_fwd_dropout == ng_dropout_mask + ng_head_fwd, bit for bit
    (the training forward draws inside the consuming launch; explicit draws stay available for the parity tests)."""
    import torch
    from nmrgnn_amd._lib import ptr
    from nmrgnn_amd.engine import Engine
    eng = Engine(make_hp(atom_feature_size=64), 10, device=gpu_device, seed=3)
    lib, h, st = eng.lib, eng.ctx.handle, eng._st()
    rng = np.random.default_rng(0)
    for n in (1, 7, 4096, 100003):
        d = torch.from_numpy(rng.uniform(0.05, 0.5, n).astype(np.float32)).to(gpu_device)
        xi = eng.randn(n, seed=77, offset=5)
        a = torch.empty(n, device=gpu_device); b = torch.empty(n, device=gpu_device)
        eng._ck(lib.ng_add_scaled(h, st, n, ptr(d), ptr(xi), 0.025, ptr(a)), "add_scaled")
        eng._ck(lib.ng_add_noise(h, st, 77, 5, n, ptr(d), 0.025, ptr(b)), "add_noise")
        assert torch.equal(a, b)
    C = 10
    for N, Fh in ((1, 32), (1000, 32), (777, 64), (2770, 128), (33, 16)):      # Fh = 16: no fast head kernel, two launches inside
        g = torch.from_numpy(rng.standard_normal((N, Fh)).astype(np.float32)).to(gpu_device)
        Wo = torch.from_numpy(rng.standard_normal((Fh, C)).astype(np.float32)).to(gpu_device)
        bo = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(gpu_device)
        at = torch.zeros(N, C, device=gpu_device); at[torch.arange(N), torch.from_numpy(rng.integers(0, C, N)).to(gpu_device)] = 1.0
        sd = torch.from_numpy(rng.uniform(0.5, 2.0, C).astype(np.float32)).to(gpu_device)
        av = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).to(gpu_device)
        m_ref = eng.dropout_mask(N * Fh, seed=91, offset=1 << 40, keep=0.8).reshape(N, Fh)
        p_ref = torch.empty(N, device=gpu_device); p = torch.empty(N, device=gpu_device)
        m = torch.full((N, Fh), -1.0, device=gpu_device)
        eng._ck(lib.ng_head_fwd(h, st, N, Fh, C, ptr(g), ptr(m_ref), ptr(Wo), ptr(bo), ptr(at), ptr(sd), ptr(av), ptr(p_ref)), "head")
        eng._ck(lib.ng_head_fwd_dropout(h, st, N, Fh, C, ptr(g), 91, 1 << 40, 0.8, ptr(m), ptr(Wo), ptr(bo), ptr(at), ptr(sd),
                                        ptr(av), ptr(p)), "head_dropout")
        assert torch.equal(m, m_ref), (N, Fh, int((m != m_ref).sum()), m[:2], m_ref[:2])
        assert torch.equal(p, p_ref), (N, Fh, float((p - p_ref).abs().max()))
