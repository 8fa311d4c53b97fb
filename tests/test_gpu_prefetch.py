"""BatchPrefetcher (nmrgnn_amd/graph.py): the next batch's device copy and lists are built on a side stream with their own
library context while the current step runs.  The lists must be the same bits as from GraphBatch on the compute stream, and a
training trajectory over a stream of different batches must not change by a bit — with big batches (the list builders'
multi-launch path and its context scratch in flight beside the step) and with single small graphs (the one-launch path)."""
import numpy as np
import pytest

from helpers import make_hp

pytestmark = pytest.mark.gpu


def _tuples(n_batches, graphs, atoms, seed):
    from nmrgnn_amd import synth
    out = []
    for i in range(n_batches):
        b = synth.make_batch(graphs, atoms + 3 * i, 16, 10, 0.1, seed=seed + i)
        out.append(b)
    return out


def _item(b):
    return (b["atoms"], b["nlist"], b["edges"], b["inv_degree"]), b["graph_ptr"]


@pytest.mark.parametrize("graphs,atoms", [(1, 40), (96, 180)])
def test_prefetched_batches_hold_the_same_lists(gpu_device, graphs, atoms):
    import torch
    from nmrgnn_amd.graph import BatchPrefetcher, GraphBatch
    bs = _tuples(4, graphs, atoms, 5)
    got = list(BatchPrefetcher([_item(b) for b in bs], device=gpu_device))
    assert len(got) == len(bs)
    for b, g in zip(bs, got):
        ref = GraphBatch(*_item(b)[0], graph_ptr=b["graph_ptr"], device=gpu_device)
        assert g._ctx is None and g.G == ref.G and g.N == ref.N
        assert torch.equal(g.nlist_c, ref.nlist_c)
        n_in = int(ref.csc()[0][-1])
        assert torch.equal(g.csc()[0], ref.csc()[0])
        assert torch.equal(g.csc()[1][:n_in], ref.csc()[1][:n_in])
        lv, lr = g.live_edges(), ref.live_edges()
        n_live = int(lr[3])
        assert int(lv[3]) == n_live
        assert torch.equal(lv[0][:n_live], lr[0][:n_live]) and torch.equal(lv[1], lr[1])
        assert torch.equal(lv[2][:n_live], lr[2][:n_live])


def test_plain_tuples_and_an_empty_source(gpu_device):
    from nmrgnn_amd.graph import BatchPrefetcher
    b = _tuples(1, 2, 30, 1)[0]
    got = list(BatchPrefetcher([_item(b)[0]], device=gpu_device, validate=False))
    assert len(got) == 1 and got[0].G == 1 and got[0].N == 60
    assert list(BatchPrefetcher([], device=gpu_device)) == []


@pytest.mark.parametrize("graphs,atoms,F", [(1, 60, 64), (64, 200, 64), (24, 120, 256)])
def test_training_over_prefetched_batches_is_bit_identical(gpu_device, graphs, atoms, F):
    import torch
    from nmrgnn_amd.engine import Engine
    from nmrgnn_amd.graph import BatchPrefetcher, GraphBatch
    from nmrgnn_amd.train import Trainer
    bs = _tuples(6, graphs, atoms, 17)
    out = {}
    for mode in ("plain", "prefetch"):
        eng = Engine(make_hp(atom_feature_size=F), 10, device=gpu_device, seed=3)
        tr = Trainer(eng, lr=1e-3)
        if mode == "plain":
            stream = (GraphBatch(*_item(b)[0], graph_ptr=b["graph_ptr"], device=gpu_device) for b in bs)
        else:
            stream = BatchPrefetcher((_item(b) for b in bs), device=gpu_device)
        losses = []
        for s, (b, gb) in enumerate(zip(bs, stream)):
            y = torch.from_numpy(b["y"]).to(gpu_device)
            w = torch.from_numpy(b["w"]).to(gpu_device)
            losses.append(tr.step(gb, y, w, seed=50 + s))
        out[mode] = ([float(x.cpu()) for x in losses], eng.params.flat.detach().cpu().numpy().copy())
    assert out["plain"][0] == out["prefetch"][0]
    assert np.array_equal(out["plain"][1], out["prefetch"][1])
