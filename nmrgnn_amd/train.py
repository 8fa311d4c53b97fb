"""Training step for the hot path: forward (noise + dropout) -> NameLoss(s=1) -> backward ->
gradient all-reduce -> Adam.  Mirrors what keras ``model.fit`` does per record in the reference
(nmrgnn/main.py:79-80 with nmrgnn/model.py:44-52), batched over many graphs."""
from __future__ import annotations

import torch

from .engine import Engine
from .graph import GraphBatch
from .parallel import GradBuckets, shard_grad_weight


class Trainer:
    """Owns the weight update of ``engine``.

    Contract on the packed weight images (``cache_images``, default True): while a Trainer is attached, the engine
    keeps the fp16-piece images of its weights across calls instead of packing them per call; ``Engine.adam_step``
    (``ng_adam_step`` inside an ``ng_weights_frozen`` window) and ``load_state_dict`` rebuild them.  Anything ELSE that
    writes into ``engine.params.flat`` in place between steps (weight surgery, finite-difference probes, a manual
    broadcast, clipping) must call ``engine.weights_changed()`` afterwards, or the next forward multiplies with the
    old images.  ``Trainer(..., cache_images=False)`` keeps per-call packing (always correct, one repack launch per
    call slower); ``close()`` restores the engine's previous setting."""

    def __init__(self, engine: Engine, lr=None, loss_balance=1.0, cache_images=True):
        self.engine = engine
        # the trainer owns the weight update: packed weight images are kept and refreshed in one launch behind Adam
        self._prev_cache_images = engine.cache_images
        engine.cache_images = bool(cache_images)
        self.lr = lr
        self.loss_balance = float(loss_balance)      # NameLoss s (build_GNNModel's loss_balance)
        P = engine.params
        # edge-MLP parameters sit first in the flat buffer (params.param_shapes)
        first_node = next(k for k in P.offsets if not k.startswith("edge_fc/"))
        self.buckets = GradBuckets(P.grad, P.offsets[first_node])
        self.step_count = 0
        # measure_comm = True: two events per step around the wait for the gradient all-reduces on the compute stream
        # (the time the collectives are EXPOSED, i.e. not hidden under the edge-MLP backward); comm_exposed_ms() reads them
        self.measure_comm = False
        self._comm_events = []

    def step(self, batch: GraphBatch, y: torch.Tensor, w: torch.Tensor, seed=None, total_graphs=None):
        """One optimiser step.  ``total_graphs``: number of graphs over ALL ranks this step (every rank can
        compute it from ``parallel.shard_range``); needed only when the shards are uneven — the loss is a mean
        over graphs, so a rank holding G_local of G_total graphs must weigh its gradient G_local/G_total, not
        1/world.  Default: equal shards."""
        eng = self.engine
        world, rank = self.buckets.world(), self.buckets.rank()
        if seed is None:
            # different noise / dropout draws on every rank and every step
            # keyed on the optimiser step, which checkpoints carry (Engine.adam_t): a resumed run continues the
            # sequence of draws instead of replaying it
            seed = 0x9E3779B97F4A7C15 ^ (eng.adam_t * 1000003) ^ (rank * 0x5851F42D4C957F2D)
        seed &= (1 << 63) - 1
        wgt = shard_grad_weight(batch.G, world, total_graphs)
        if self.loss_balance == 1.0:
            # the L2 loss is taken inside the forward: head, loss and the head's backward are one launch where the shape allows
            eng.forward(batch, training=True, seed=seed, loss=(y, w, wgt))
            loss, dpred = eng.tape.loss, None
        else:
            peaks = eng.forward(batch, training=True, seed=seed)
            loss, dpred = eng.loss_name(batch, y, w, peaks, self.loss_balance)
            if wgt != 1.0:
                dpred.mul_(wgt)
        eng.backward(dpred, on_node_grads=self.buckets.launch_node)
        self.buckets.launch_edge()
        if self.measure_comm:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.buckets.wait()
            e1.record()
            self._comm_events.append((e0, e1))
        else:
            self.buckets.wait()
        eng.adam_step(lr=self.lr, grad_scale=self.buckets.grad_scale())
        self.step_count += 1
        return loss

    def comm_exposed_ms(self):
        """mean stall of the compute stream at the all-reduce wait over the steps measured so far (resets)"""
        if not self._comm_events:
            return 0.0
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._comm_events]
        self._comm_events = []
        return sum(ms) / len(ms)

    def close(self):
        """Detach from the engine: restore its previous ``cache_images`` setting (a bare Engine packs per call)."""
        self.engine.cache_images = self._prev_cache_images
        if not self.engine.cache_images:
            self.engine.weights_changed()
