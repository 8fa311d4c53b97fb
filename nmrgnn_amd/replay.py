"""Graph replay of small calls: the reference's own call granularity at the GPU's speed.

The reference trains on ONE graph per step (nmrgnn/library.py:88-89: ``dataset`` yields one record, keras ``fit`` steps on
it, nmrgnn/main.py:74-80) and predicts ONE structure per call (main.py:236-245).  At that size a step is ~33 launches of a
few microseconds each and a frame ~8: the host's launch rate, not the GPU, sets the time.  Here the chain of library calls
is captured ONCE per shape as a HIP graph (``torch.cuda.CUDAGraph`` on the stream the engine passes to every call) and
replayed with one graph launch per step.

What makes the chain capturable (include/nmrgnn_hip.h, "graph replay of small calls"): every entry point is asynchronous
and allocates nothing once the context's scratch is sized — a warm-up step before the capture sizes it; the tensors the
engine allocates per call come from the graph's private pool of the caching allocator.  What changes from step to step
are three launch arguments (noise seed, dropout seed, Adam's bias-corrected rate) and the inputs: ``ng_replay_stage``, ONE
eager launch per step, writes the former into a device block the armed context's kernels read, and copies the latter into
the static buffers the captured chain was recorded on.  After every replayed training step ``ng_replay_commit`` does the host
side of the captured ``ng_adam_step`` (weight version, which packed images are current): eager calls in between replays —
a validation batch, a step of another shape — see the weights of NOW.

Results are the bits of the eager chain (tests/test_gpu_replay.py: parameter trajectories and peaks ``torch.equal``)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import ptr
from .graph import GraphBatch, frames_to_batch


def _stage(eng, seed, lr, step, pairs, beta1=0.9, beta2=0.999):
    """ng_replay_stage: per-step state + device-to-device copies ``pairs`` = [(src tensor, static dst tensor), ...]"""
    n = len(pairs)
    if n > 8:
        raise ValueError("ng_replay_stage copies at most 8 buffers")
    src = (C.c_void_p * max(n, 1))(*[p[0].data_ptr() for p in pairs])
    dst = (C.c_void_p * max(n, 1))(*[p[1].data_ptr() for p in pairs])
    nb = (C.c_uint64 * max(n, 1))(*[p[1].numel() * p[1].element_size() for p in pairs])
    eng._ck(eng.lib.ng_replay_stage(eng.ctx.handle, eng._st(), C.c_uint64(seed & ((1 << 64) - 1)), float(lr), beta1, beta2,
                                    int(step), n, src, dst, nb), "ng_replay_stage")


def _like(dst, x):
    """``x`` as a contiguous device tensor of dst's dtype and shape (a view when it already is one: no launch)"""
    x = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    x = x.to(device=dst.device, dtype=dst.dtype)
    if x.shape != dst.shape:
        x = x.reshape(dst.shape)
    return x.contiguous()


class TrainStepReplay:
    """``Trainer.step`` for batches of ONE fixed shape (N atoms, K neighbour slots, the graph partition ``graph_ptr``),
    replayed as a HIP graph.  ``step(tuple, y, w)`` = ``trainer.step(GraphBatch(*tuple, graph_ptr=...), y, w)`` bit for
    bit: same noise / dropout draws (the seed sequence of Trainer.step), same Adam step count.

    Single process (the gradient all-reduce of a data-parallel world is not captured).  The loss tensor returned by
    ``step`` is a static buffer: read it before the next step."""

    def __init__(self, trainer, example, y, w, graph_ptr=None):
        eng = trainer.engine
        if trainer.buckets.world() != 1:
            raise ValueError("TrainStepReplay: single-process training only")
        if getattr(trainer, "measure_comm", False):
            raise ValueError("TrainStepReplay: measure_comm records timing events, which a captured step cannot hold")
        self.trainer, self.eng = trainer, eng
        dev = eng.device
        atoms, nlist, edges, inv = example
        as_t = lambda x, dt: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))).to(device=dev, dtype=dt).contiguous()
        self.s_atoms = as_t(atoms, torch.float32).clone()
        self.s_nlist = as_t(nlist, torch.int32).clone()
        self.s_edges = as_t(edges, torch.float32).clone()
        self.s_inv = as_t(inv, torch.float32).reshape(-1).clone()
        self.s_y = as_t(y, torch.float32).clone()
        self.s_w = as_t(w, torch.float32).clone()
        N = self.s_atoms.shape[0]
        self.graph_ptr = np.asarray([0, N] if graph_ptr is None else graph_ptr, dtype=np.int32)
        self.lr = trainer.lr if trainer.lr is not None else float(eng.hp.get('learning_rate'))
        self._stream = torch.cuda.Stream(device=dev)
        self._graph = torch.cuda.CUDAGraph()
        self.loss = None
        self._capture()

    def _one_step(self):
        gb = GraphBatch(self.s_atoms, self.s_nlist, self.s_edges, self.s_inv, graph_ptr=self.graph_ptr, device=self.eng.device,
                        validate=False)
        # the captured chain reads gb's device-side graph boundaries (and whatever else the batch allocated outside the
        # graph's pool) at every replay: the batch lives as long as the replay object, not as long as a cache entry
        self._gb = gb
        return self.trainer.step(gb, self.s_y, self.s_w)

    def _capture(self):
        eng, tr = self.eng, self.trainer
        P = eng.params
        # Two warm-up steps size every scratch buffer of the context and fill the caches the step relies on (graph_ptr on the
        # device, packed weight images, the reduction queue).  They run with a staged rate of ZERO: the parameters — and with
        # them every packed image the captured chain will read — stay as they are bit for bit; Adam's moments, which the
        # warm-up does change, and the host-side step counters (the capture itself runs no kernel but counts steps) are
        # put back afterwards.
        keep = (eng.adam_m.clone(), eng.adam_v.clone(), eng.adam_t, tr.step_count)
        cur = torch.cuda.current_stream(eng.device)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            eng._ck(eng.lib.ng_replay_arm(eng.ctx.handle, 1), "ng_replay_arm")
            try:
                for _ in range(2):
                    _stage(eng, 1, 0.0, 1, [])
                    self._one_step()
                torch.cuda.synchronize(eng.device)
                with torch.cuda.graph(self._graph, stream=self._stream):
                    self.loss = self._one_step()
                tok = C.c_uint64(0)
                eng._ck(eng.lib.ng_replay_token(eng.ctx.handle, C.byref(tok)), "ng_replay_token")
                self._token = int(tok.value)
            finally:
                eng.lib.ng_replay_arm(eng.ctx.handle, 0)
            eng.adam_m.copy_(keep[0]); eng.adam_v.copy_(keep[1])
            eng.adam_t, tr.step_count = keep[2], keep[3]
        cur.wait_stream(self._stream)

    def step(self, graph_tuple, y, w):
        eng, tr = self.eng, self.trainer
        atoms, nlist, edges, inv = graph_tuple
        # the seed sequence of Trainer.step (rank 0)
        seed = (0x9E3779B97F4A7C15 ^ (eng.adam_t * 1000003)) & ((1 << 63) - 1)
        pairs = [(_like(self.s_atoms, atoms), self.s_atoms), (_like(self.s_nlist, nlist), self.s_nlist),
                 (_like(self.s_edges, edges), self.s_edges), (_like(self.s_inv, inv), self.s_inv),
                 (_like(self.s_y, y), self.s_y), (_like(self.s_w, w), self.s_w)]
        eng._ck(eng.lib.ng_replay_arm(eng.ctx.handle, 1), "ng_replay_arm")
        try:
            _stage(eng, seed, self.lr, eng.adam_t + 1, pairs)
            self._graph.replay()
            # the host bookkeeping of the replayed ng_adam_step: new weight version, and only the images the captured launch
            # rebuilds are current (a cached image of any other call must not be served with the weights of N steps ago)
            eng._ck(eng.lib.ng_replay_commit(eng.ctx.handle, C.c_uint64(self._token)), "ng_replay_commit")
        finally:
            eng.lib.ng_replay_arm(eng.ctx.handle, 0)
        eng.adam_t += 1
        tr.step_count += 1
        return self.loss


class ForwardReplay:
    """``engine.forward(frames_to_batch(atoms, positions, K))`` for ONE frame shape — the kNN graph build on the GPU plus
    the model forward of a structure (``eval-struct``'s inner loop, nmrgnn/main.py:236-245) — replayed as a HIP graph.
    ``__call__(positions[n,3])`` returns the static ``peaks[n]`` tensor (read it before the next call)."""

    def __init__(self, engine, atoms, positions, neighbor_number=16):
        self.eng = engine
        dev = engine.device
        self.atoms = (atoms if isinstance(atoms, torch.Tensor) else torch.as_tensor(np.asarray(atoms))).to(device=dev, dtype=torch.float32)
        pos = (positions if isinstance(positions, torch.Tensor) else torch.as_tensor(np.asarray(positions))).to(device=dev, dtype=torch.float32)
        self.s_pos = pos.reshape(1, -1, 3).contiguous().clone()
        self.K = int(neighbor_number)
        self._stream = torch.cuda.Stream(device=dev)
        self._graph = torch.cuda.CUDAGraph()
        cur = torch.cuda.current_stream(dev)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            for _ in range(2):
                self._one()
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(self._graph, stream=self._stream):
                self.peaks = self._one()
        cur.wait_stream(self._stream)

    def _one(self):
        self._gb = frames_to_batch(self.atoms, self.s_pos, self.K, device=self.eng.device)   # kept alive: the chain reads its buffers
        out = self.eng.forward(self._gb)
        self._table = self.eng._table_cache      # (the edge-function table of frozen weights the captured calls read)
        return out

    def __call__(self, positions):
        p = _like(self.s_pos, positions)
        if p.data_ptr() != self.s_pos.data_ptr():
            _stage(self.eng, 0, 1e-4, 1, [(p, self.s_pos)])
        self._graph.replay()
        return self.peaks
