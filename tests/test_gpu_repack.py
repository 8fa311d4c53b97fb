"""Weight images refreshed in ONE launch behind ng_adam_step (round 4; csrc/repack.hip, pack_bodies.cuh) against the
per-call packing of rounds 1-3: the images are the same bits whichever launch builds them, so whole training
trajectories must agree bit for bit — also when a weight leaves the fp16 piece range on the way (the image's flag word
then sends the window kernels to their fp32-input bodies), and when two models take turns on one device."""
import numpy as np
import pytest

from helpers import make_hp, small_batch

pytestmark = pytest.mark.gpu


def _setup(dev, F, seed=3, cache=True):
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import GraphBatch
    from nmrgnn_amd.train import Trainer
    b = small_batch(n_graphs=4, n_atoms=70, seed=11)
    eng = Engine(make_hp(atom_feature_size=F), 10, device=dev, seed=seed)
    gb = GraphBatch(b["atoms"], b["nlist"], b["edges"], b["inv_degree"], graph_ptr=b["graph_ptr"], device=dev)
    tr = Trainer(eng, lr=1e-3)
    eng.cache_images = cache
    y = torch.from_numpy(b["y"]).to(dev)
    w = torch.from_numpy(b["w"]).to(dev)
    return eng, gb, tr, y, w


def _run(tr, gb, y, w, steps, seed0=100):
    losses = []
    for s in range(steps):
        losses.append(float(tr.step(gb, y, w, seed=seed0 + s).cpu()))
    return losses, tr.engine.params.flat.detach().cpu().numpy().copy()


@pytest.mark.parametrize("F", [64, 256])
def test_refreshed_images_give_the_trajectory_of_per_call_packing(gpu_device, F):
    out = {}
    for cache in (False, True):
        eng, gb, tr, y, w = _setup(gpu_device, F, cache=cache)
        out[cache] = _run(tr, gb, y, w, 5)
    assert out[False][0] == out[True][0]
    assert np.array_equal(out[False][1], out[True][1])


def test_a_weight_leaving_the_piece_range_mid_training(gpu_device):
    """after two steps one MPLayer weight is set to 400 (2^8 * 400 > 65504: no fp16 pieces) through load_state_dict; the
    following steps refresh the images behind Adam with the flag word raised; then the weight comes back into range"""
    out = {}
    for cache in (False, True):
        eng, gb, tr, y, w = _setup(gpu_device, 64, cache=cache)
        _run(tr, gb, y, w, 2)
        sd = eng.params.state_dict()
        sd["mp/1/w"][3, 5, 1] = 400.0
        eng.params.load_state_dict(sd)
        l1, p1 = _run(tr, gb, y, w, 3, seed0=200)
        sd = eng.params.state_dict()
        sd["mp/1/w"][3, 5, 1] = 0.25
        eng.params.load_state_dict(sd)
        l2, p2 = _run(tr, gb, y, w, 2, seed0=300)
        out[cache] = (l1, p1, l2, p2)
        assert np.all(np.isfinite(p1)) and np.all(np.isfinite(p2))
    assert out[False][0] == out[True][0] and out[False][2] == out[True][2]
    assert np.array_equal(out[False][1], out[True][1])
    assert np.array_equal(out[False][3], out[True][3])


def test_two_models_taking_turns(gpu_device):
    """the image cache belongs to one model at a time (ng_weights_frozen owner): alternating trainers must each follow the
    trajectory they follow alone"""
    solo = []
    for seed in (3, 4):
        eng, gb, tr, y, w = _setup(gpu_device, 64, seed=seed)
        solo.append(_run(tr, gb, y, w, 4)[1])
    a = _setup(gpu_device, 64, seed=3)
    b = _setup(gpu_device, 64, seed=4)
    for s in range(4):
        a[2].step(a[1], a[3], a[4], seed=100 + s)
        b[2].step(b[1], b[3], b[4], seed=100 + s)
    assert np.array_equal(a[0].params.flat.cpu().numpy(), solo[0])
    assert np.array_equal(b[0].params.flat.cpu().numpy(), solo[1])


def test_inference_between_steps_sees_the_updated_weights(gpu_device):
    eng, gb, tr, y, w = _setup(gpu_device, 64)
    ref_eng, _, ref_tr, _, _ = _setup(gpu_device, 64, cache=False)
    for s in range(3):
        tr.step(gb, y, w, seed=s)
        ref_tr.step(gb, y, w, seed=s)
        assert np.array_equal(eng.forward(gb).cpu().numpy(), ref_eng.forward(gb).cpu().numpy())
