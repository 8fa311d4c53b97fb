"""The hot-path engine: forward and hand-derived backward of GNNModel.call
(nmrgnn/model.py:245-274) as a sequence of C-ABI calls into libnmrgnn_hip.so.

torch is used for device memory and streams only; every arithmetic step is a HIP kernel.
There is no CPU fallback — constructing an Engine without the library / a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import ptr, ptr_array
from .graph import GraphBatch
from .params import ParamStore

ACT = {None: 0, "linear": 0, "softplus": 1, "relu": 2, "tanh": 3}
DROPOUT_RATE = 0.2   # nmrgnn/model.py:217
# batches up to this many atoms take the fused FC-block + head launch in inference (above it the per-layer GEMMs have
# enough rows to fill the chip and the launch chain no longer sets the pace)
FUSED_TAIL_MAX_ATOMS = 16384


def rbf_grid(low, high, count):
    """nmrgnn/layers.py:126-129 — float32 linspace as tf.linspace computes it, gap = c[1]-c[0]."""
    lo, hi = np.float32(low), np.float32(high)
    delta = (hi - lo) / np.float32(count - 1)
    c = (lo + delta * np.arange(count, dtype=np.float32)).astype(np.float32)
    c[-1] = hi
    return c, float(np.float32(c[1] - c[0]))


class Tape:
    """activations kept between forward(training=True) and backward()"""
    __slots__ = ("batch", "d_eff", "z_save", "z_layout", "e", "h", "A", "S", "fx", "fs", "g", "drop_mask",
                 "peaks", "live", "table", "table_sync", "loss", "dpeaks", "dg", "head_partial")


class Engine:
    _ids = 0

    def __init__(self, hp, num_elem, peak_std=None, peak_avg=None, device=None, seed=1234):
        if not torch.cuda.is_available():
            raise _lib.NGError("nmrgnn_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.ctx = _lib.get_context(self.device.index)
        self.lib = self.ctx.lib
        self.hp = hp
        self.C = int(num_elem)
        self.F = hp.get('atom_feature_size')
        self.E = hp.get('edge_feature_size')
        self.H = hp.get('edge_hidden_size')
        self.L = hp.get('mp_layers')
        self.Le = hp.get('edge_fc_layers')
        self.Lf = hp.get('fc_layers')
        self.sigma = float(hp.get('noise'))
        self.use_dropout = bool(hp.get('dropout'))
        if hp.get('fc_activation') not in ACT or hp.get('mp_activation') not in ACT:
            raise ValueError(f"unsupported activation {hp.get('fc_activation')!r} / {hp.get('mp_activation')!r}")
        self.fc_act = ACT[hp.get('fc_activation')]
        self.mp_act = ACT[hp.get('mp_activation')]
        self.params = ParamStore(hp, self.C, self.device, seed=seed)
        self.params.on_change = self._on_params_changed
        c, gap = rbf_grid(hp.get('rbf_low'), hp.get('rbf_high'), self.H)
        self.centers = torch.from_numpy(c).to(self.device)
        self.gap = gap
        std = np.ones(self.C, np.float32) if peak_std is None else np.asarray(peak_std, np.float32)[:self.C]
        avg = np.zeros(self.C, np.float32) if peak_avg is None else np.asarray(peak_avg, np.float32)[:self.C]
        self.peak_std = torch.from_numpy(np.ascontiguousarray(std)).to(self.device)
        self.peak_avg = torch.from_numpy(np.ascontiguousarray(avg)).to(self.device)
        # Adam state (keras defaults, nmrgnn/model.py:44-45)
        self.adam_m = torch.zeros_like(self.params.flat)
        self.adam_v = torch.zeros_like(self.params.flat)
        self.adam_t = 0
        self.tape = None
        self._rng_calls = 0
        self._frozen = False
        # training with refreshed images (round 4): the library keeps every packed weight image of this engine and
        # ng_adam_step rebuilds them all in ONE launch right behind the update, instead of one small launch in front of
        # every consumer of the next step.  Weights then have to change through adam_step / params.load_state_dict (or be
        # followed by weights_changed()) — the Trainer, which owns the update, switches it on; a bare Engine keeps packing
        # per call, so code that writes into parameter views directly stays correct.
        self.cache_images = False
        self.defer_reductions = True      # backward(): queue the weight-gradient sums, one launch (ng_defer_reductions)
        # edge shapes without the fused live-edge kernels (edge_hidden_size != 128, ...): the same table, its guard read on the
        # HOST (one synchronisation per call: nothing on the device can gate the layered kernels) — False / NG_EDGE_TABLE_SYNC=0: per edge
        self.edge_table_sync = os.environ.get("NG_EDGE_TABLE_SYNC", "1") != "0"
        # forward(loss=...): head + L2 loss + head backward as one launch where the shape allows (ng_head_loss_bwd)
        self.fuse_head_loss = os.environ.get("NG_HEAD_LOSS", "1") != "0"
        # padded slots (edges == 0) are skipped by the fused edge kernels (include/nmrgnn_hip.h: ng_edge_mlp_fwd_live);
        # NG_EDGE_LIVE=0 runs every slot as rounds 1-3 did (A/B measurements, tests)
        self.use_live_edges = os.environ.get("NG_EDGE_LIVE", "1") != "0"
        # DEFAULT since round 6 (round 5: opt-in; csrc/edge_table.hip): the edge MLP is a function of one scalar per edge —
        # evaluate it with the fused kernels on EDGE_TABLE_POINTS equidistant distances and interpolate every edge (cubic, error
        # < 1e-9 relative for weights of ordinary size); the backward scatters de onto the table (exact adjoint) and runs the
        # fused backward on the table.  Guarded on the device: the same launch evaluates the function at the midpoints too,
        # ng_edge_table_check compares, and above ``edge_table_tol`` x max |e| the SAME call is answered by the per-edge
        # kernels (they run over a row count that is zero unless the guard is up).  NG_EDGE_TABLE=0 / edge_table = False: the
        # per-edge kernels always — what the reference does and what bench.py's `value` is measured on.
        self.edge_table = os.environ.get("NG_EDGE_TABLE", "1") != "0"
        self.edge_table_tol = 3.0e-6          # bound on |interpolant - function| at the midpoints, relative to max |e| of the table
        self.edge_table_min_edges = 262144    # below, the table's extra launches (range, check, gated per-edge launch, interpolation:
                                              # ~60 us) cost more than the per-edge MLP itself (7lgi, one 44 K-edge frame per call: 0.161 against 0.154 ms)
        self.edge_table_force_fallback = False   # tests: raise the guard whatever the check finds (tol = -1)
        self._wgen = 0                        # bumped whenever the weights change: a table kept over calls (frozen weights)
        self._table_cache = None
        Engine._ids += 1
        self._id = Engine._ids          # owner tag of the frozen-weight cache

    # ------------------------------------------------------------------ frozen weights (inference)
    def freeze_weights(self, on=True):
        """Inference with constant weights: the library keeps its packed weight images across calls instead of
        re-packing them on every call (several launches per call at molecule size).  Weight changes through
        ``params.load_state_dict`` / ``adam_step`` are noticed; after writing into a parameter view directly call
        ``weights_changed()``.  The library's cache is keyed by weight addresses and shared by everything on the
        device, so it is switched on only for the duration of THIS engine's forward calls, under this engine's id."""
        self._frozen = bool(on)
        if not on:
            self._ck(self.lib.ng_weights_frozen(self.ctx.handle, 0), "ng_weights_frozen")

    def weights_changed(self):
        self._wgen += 1
        self._ck(self.lib.ng_weights_changed(self.ctx.handle), "ng_weights_changed")

    def _on_params_changed(self):
        self._wgen += 1
        self.lib.ng_weights_changed(self.ctx.handle)

    # ------------------------------------------------------------------ helpers
    def _st(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _new(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def _ck(self, rc, what):
        self.ctx.check(rc, what)

    def randn(self, n, seed, offset=0):
        out = self._new(n)
        self._ck(self.lib.ng_randn(self.ctx.handle, self._st(), seed, offset, ptr(out), n), "ng_randn")
        return out

    def dropout_mask(self, n, seed, offset=0, keep=1.0 - DROPOUT_RATE):
        out = self._new(n)
        self._ck(self.lib.ng_dropout_mask(self.ctx.handle, self._st(), seed, offset, keep, ptr(out), n),
                 "ng_dropout_mask")
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, batch: GraphBatch, training=False, noise=None, dropout_mask=None, seed=0, keep_tape=None, loss=None):
        """peaks[N].  training=True keeps the tape for backward(); ``keep_tape=True`` keeps it for an inference-mode
        forward too (no noise, no dropout — what autograd through ``model(g, training=False)`` differentiates).
        ``noise`` (xi[N,K], standard normal) / ``dropout_mask`` ([N,F/2], values 0 or 1/keep) may be
        supplied explicitly (parity tests); otherwise they are drawn on the GPU from ``seed``.
        ``loss = (y, w, grad_weight)``: the L2 NameLoss (``loss_l2``) of the peaks is taken inside the forward —
        ``tape.loss`` holds it ONCE ``backward(None)`` HAS RUN (which continues from its gradient); where the shape
        allows, the head, the loss and the head's backward are ONE launch whose second stage rides with the backward's
        other reductions (``ng_head_loss_bwd``; ``fuse_head_loss = False`` / NG_HEAD_LOSS=0: three calls)."""
        if loss is not None and not (training if keep_tape is None else keep_tape):
            raise ValueError("forward(loss=...) needs a tape (training=True)")
        if not (self._frozen or self.cache_images):
            return self._forward(batch, training, noise, dropout_mask, seed, keep_tape, loss)
        self._ck(self.lib.ng_weights_frozen(self.ctx.handle, self._id), "ng_weights_frozen")
        try:
            return self._forward(batch, training, noise, dropout_mask, seed, keep_tape, loss)
        finally:
            self.lib.ng_weights_frozen(self.ctx.handle, 0)

    def _forward(self, batch, training, noise, dropout_mask, seed, keep_tape=None, loss=None):
        tape = bool(training) if keep_tape is None else bool(keep_tape)
        lib, h, st = self.lib, self.ctx.handle, self._st()
        P = self.params
        N, K, F, E, H = batch.N, batch.K, self.F, self.E, self.H
        if batch.C != self.C:
            raise ValueError(f"atoms has {batch.C} element columns, model was built for {self.C}")
        ne = batch.n_edges
        # locality hint for the default-width aggregation (results do not depend on it)
        self._ck(lib.ng_ctx_set_graph_span(h, batch.max_graph_atoms), "ng_ctx_set_graph_span")
        d_src = batch.edges.reshape(-1)
        d_eff = d_src
        # live-edge view: the fused edge kernels walk the compacted live slots only (same e bit for bit)
        # (the live kernels index e with 32 bits: n_slots * E * 4 < 2^31, check_live in edge_fwd_h2.hip; larger batches take the
        # every-slot kernels)
        live_ok = ne * E * 4 < (1 << 31) and bool(lib.ng_edge_live_supported(H, E, self.Le, self.fc_act))
        use_table = self.edge_table and E <= 8 and ne >= self.edge_table_min_edges and live_ok
        live = batch.live_edges(force=use_table) if (live_ok and (use_table or self.use_live_edges)) else None
        if live is not None:
            perm, pos, d_c, n_live = live
            d_src = d_eff = d_c
            if training and self.sigma > 0:
                d_eff = self._new(ne)       # compacted: the first n_live entries are used
                self._ck(lib.ng_add_noise_live(h, st, seed, 0, ne, ptr(batch.edges), ptr(None if noise is None else noise.reshape(-1)),
                                               self.sigma, ptr(pos), ptr(d_eff)), "ng_add_noise_live")
        elif training and self.sigma > 0:
            d_eff = self._new(ne)
            if noise is None:     # the draw and d + sigma * xi in one launch (the bits of randn + add_scaled)
                self._ck(lib.ng_add_noise(h, st, seed, 0, ne, ptr(d_src), self.sigma, ptr(d_eff)), "ng_add_noise")
            else:
                noise = noise.reshape(-1)
                self._ck(lib.ng_add_scaled(h, st, ne, ptr(d_src), ptr(noise), self.sigma, ptr(d_eff)),
                         "ng_add_scaled")
        # the table of the edge function and the guard's verdict (device words) — before the per-edge launch, which runs over
        # gate[1] rows: none unless the guard is up
        table = self._edge_table_build(batch, live, d_eff, tape, training) if (use_table and live is not None) else None
        table_sync = None
        if (table is None and live is None and self.edge_table and self.edge_table_sync and not live_ok and not batch.is_csr
                and E <= 8 and ne >= self.edge_table_min_edges and ne * E * 4 < (1 << 31)
                and not torch.cuda.is_current_stream_capturing()):
            table_sync = self._edge_table_build_sync(batch, d_src, d_eff, tape)      # None: the guard is up
        if table_sync is not None:
            z_save, z_layout = None, 0
            e = self._new(ne, E)
            self._ck(lib.ng_edge_table_interp(h, st, ne, E, table_sync["T"], ptr(batch.edges), ptr(d_eff), None,
                                              ptr(table_sync["rng"]), ptr(table_sync["e_all"]), ptr(table_sync["gate"]), ptr(e)),
                     "ng_edge_table_interp")
        z_save = self._new(self.Le - 1, ne, H) if (tape and table_sync is None) else None
        # element order of the tape the edge forward is about to write (it depends on the NG_EDGE_* switches in force
        # NOW; the backward is told, so a switch flipped in between cannot make it misread the tape)
        z_layout = int(lib.ng_edge_tape_layout(H, E, self.Le, self.fc_act, ne)) if (tape and table_sync is None) else 0
        W = [P[f"edge_fc/{t}/kernel"] for t in range(self.Le)]
        B = [P[f"edge_fc/{t}/bias"] for t in range(self.Le)]
        if table_sync is not None:
            pass            # e came from the table
        elif live is not None:
            e = self._new(ne, E)
            rows = table["gate"][1:] if table is not None else n_live
            self._ck(lib.ng_edge_mlp_fwd_live(h, st, ne, H, E, self.Le, self.fc_act, ptr(d_src), ptr(d_eff), ptr(perm),
                                              ptr(rows), ptr(self.centers), self.gap, ptr_array(W), ptr_array(B),
                                              ptr(e), ptr(z_save)), "ng_edge_mlp_fwd_live")
            if table is not None:       # every slot from the table — unless the guard is up: then the launch returns at once
                self._ck(lib.ng_edge_table_interp(h, st, ne, E, table["T"], ptr(batch.edges), ptr(d_eff), ptr(pos),
                                                  ptr(table["rng"]), ptr(table["e_all"]), ptr(table["gate"]), ptr(e)),
                         "ng_edge_table_interp")
        else:
            e = self._new(ne, E)
            self._ck(lib.ng_edge_mlp_fwd(h, st, ne, H, E, self.Le, self.fc_act, ptr(d_src), ptr(d_eff),
                                         ptr(self.centers), self.gap, ptr_array(W), ptr_array(B),
                                         ptr(e), ptr(z_save)), "ng_edge_mlp_fwd")
        h0 = self._new(N, F)
        self._ck(lib.ng_embed_fwd(h, st, N, self.C, F, ptr(batch.atoms), ptr(P["embed/kernel"]),
                                  ptr(h0)), "ng_embed_fwd")
        hs, As, Ss = [h0], [], []
        for l in range(self.L):
            hn = self._new(N, F)
            S = self._new(N, F) if (tape and self.mp_act != 0) else None
            if batch.is_csr and not tape and N <= FUSED_TAIL_MAX_ATOMS and lib.ng_mp_layer_short_ok(N, 1, F, E):
                A = None        # molecule-sized inference over CSR lists: one launch per layer
                self._ck(lib.ng_mp_layer_fwd_short_csr(h, st, N, F, E, self.mp_act, 1, ptr(hs[-1]), ptr(batch.row_ptr),
                                                       ptr(batch.nlist), ptr(e), ptr(batch.inv_degree), ptr(P[f"mp/{l}/w"]),
                                                       ptr(hn)), "ng_mp_layer_fwd_short_csr")
            elif batch.is_csr:        # variable-degree lists: row_ptr / col instead of [N,K]
                A = self._new(N, E, F) if tape else None
                self._ck(lib.ng_mp_layer_fwd_csr(h, st, N, ne, F, E, self.mp_act, 1, ptr(hs[-1]),
                                                 ptr(batch.row_ptr), ptr(batch.nlist), ptr(e),
                                                 ptr(batch.inv_degree), ptr(P[f"mp/{l}/w"]), ptr(hn), ptr(A),
                                                 ptr(S)), "ng_mp_layer_fwd_csr")
            elif not tape and N <= FUSED_TAIL_MAX_ATOMS and lib.ng_mp_layer_short_ok(N, K, F, E):
                # molecule-sized inference: aggregate + update of the layer in ONE launch (csrc/frame_fused.hip)
                A = None
                self._ck(lib.ng_mp_layer_fwd_short(h, st, N, K, F, E, self.mp_act, 1, ptr(hs[-1]), ptr(batch.nlist_c), ptr(e),
                                                   ptr(batch.inv_degree), ptr(P[f"mp/{l}/w"]), ptr(hn)), "ng_mp_layer_fwd_short")
            else:
                A = self._new(N, E, F) if (tape and lib.ng_mp_layer_wants_aggregate(F, E, K)) else None
                self._ck(lib.ng_mp_layer_fwd(h, st, N, K, F, E, self.mp_act, 1, ptr(hs[-1]),
                                             ptr(batch.nlist_c), ptr(e), ptr(batch.inv_degree),
                                             ptr(P[f"mp/{l}/w"]), ptr(hn), ptr(A), ptr(S)),
                         "ng_mp_layer_fwd")
            hs.append(hn)
            As.append(A)
            Ss.append(S)
        Wfc = [P[f"fc/{t}/kernel"] for t in range(self.Lf)]
        Bfc = [P[f"fc/{t}/bias"] for t in range(self.Lf)]
        # molecule-sized inference: FC block + head in ONE launch (csrc/frame_fused.hip)
        # (not for a training-mode forward without a tape: the fused tail has no dropout, the layered path applies it)
        if not tape and not (training and self.use_dropout) and N <= FUSED_TAIL_MAX_ATOMS \
                and lib.ng_fc_head_ok(N, F, self.Lf, self.C, self.fc_act):
            peaks = self._new(N)
            self._ck(lib.ng_fc_head_fwd(h, st, N, F, self.Lf, self.C, self.fc_act, ptr(hs[-1]), ptr_array(Wfc), ptr_array(Bfc),
                                        ptr(P["out/kernel"]), ptr(P["out/bias"]), ptr(batch.atoms), ptr(self.peak_std),
                                        ptr(self.peak_avg), ptr(peaks)), "ng_fc_head_fwd")
            return peaks
        # FC block: every layer in one call; the tape keeps the layer inputs only (nmrgnn/model.py:191-196)
        fx, fs = [hs[-1]], []
        for t in range(self.Lf - 1):
            fx.append(self._new(N, F))
        Fh = F // 2
        g = self._new(N, Fh)
        self._ck(lib.ng_fc_block_fwd(h, st, N, F, self.Lf, self.fc_act, ptr(fx[0]), ptr_array(Wfc),
                                     ptr_array(Bfc), ptr_array(fx[1:]), ptr(g)), "ng_fc_block_fwd")
        mask = None
        peaks = self._new(N)
        loss_t = dpeaks = dg = head_partial = None
        nb = 0
        if loss is not None and self.fuse_head_loss and dropout_mask is None:
            nb = int(lib.ng_head_loss_blocks(h, batch.G, Fh, self.C, batch.max_graph_atoms))
        if nb > 0:
            y, w, gw = loss
            keep = (1.0 - DROPOUT_RATE) if (training and self.use_dropout) else 1.0
            loss_t, dg = self._new(1), self._new(N, Fh)
            head_partial = self._new(nb, Fh * self.C + self.C + 1)
            self._ck(lib.ng_head_loss_bwd(h, st, N, batch.G, Fh, self.C, batch.max_graph_atoms, ptr(g), seed, 1 << 40, keep,
                                          None, ptr(P["out/kernel"]), ptr(P["out/bias"]), ptr(batch.atoms),
                                          ptr(self.peak_std), ptr(self.peak_avg), ptr(batch.graph_ptr), ptr(y), ptr(w),
                                          float(gw), ptr(peaks), ptr(dg), ptr(head_partial)),
                     "ng_head_loss_bwd")
        elif training and self.use_dropout and dropout_mask is None:
            # the keep-mask is drawn inside the head launch and kept for the backward (the values of ng_dropout_mask)
            mask = self._new(N, Fh)
            self._ck(lib.ng_head_fwd_dropout(h, st, N, Fh, self.C, ptr(g), seed, 1 << 40, 1.0 - DROPOUT_RATE, ptr(mask),
                                             ptr(P["out/kernel"]), ptr(P["out/bias"]), ptr(batch.atoms),
                                             ptr(self.peak_std), ptr(self.peak_avg), ptr(peaks)), "ng_head_fwd_dropout")
        else:
            if training and self.use_dropout:
                mask = dropout_mask.reshape(N, Fh).contiguous()
            self._ck(lib.ng_head_fwd(h, st, N, Fh, self.C, ptr(g), ptr(mask), ptr(P["out/kernel"]),
                                     ptr(P["out/bias"]), ptr(batch.atoms), ptr(self.peak_std),
                                     ptr(self.peak_avg), ptr(peaks)), "ng_head_fwd")
        if loss is not None and nb == 0:
            y, w, gw = loss
            loss_t, dpeaks = self.loss_l2(batch, y, w, peaks)
            if gw != 1.0:
                dpeaks.mul_(gw)
        if tape:
            tp = Tape()
            tp.loss, tp.dpeaks, tp.dg, tp.head_partial = loss_t, dpeaks, dg, head_partial
            tp.batch, tp.d_eff, tp.z_save, tp.e = batch, d_eff, z_save, e
            tp.z_layout = z_layout
            tp.live = live
            tp.table = table
            tp.table_sync = table_sync
            tp.h, tp.A, tp.S, tp.fx, tp.fs, tp.g, tp.drop_mask, tp.peaks = hs, As, Ss, fx, fs, g, mask, peaks
            self.tape = tp
        return peaks

    # ------------------------------------------------------------------ edge function table (opt-in)
    EDGE_TABLE_POINTS = 4096      # E <= 4; 2048 for E = 5 .. 8 (the scatter's 64-bit table and the guard's staged copy share the LDS)

    def _table_points(self):
        return self.EDGE_TABLE_POINTS if self.E <= 4 else self.EDGE_TABLE_POINTS // 2

    def _edge_table_build(self, batch, live, d_eff_c, tape, training):
        """The table of the edge function for this call (csrc/edge_table.hip) and the guard's gate words.
        ``d_eff_c``: the distances fed to the RBF in the COMPACTED row order of the live view (noise already added when
        training).  With frozen weights and no tape the table of an earlier call is reused while the weights stay the same:
        then only the call's distance range is checked against the table's."""
        lib, h, st = self.lib, self.ctx.handle, self._st()
        P = self.params
        perm, pos, d_c, n_live = live
        ne, E, H, T = batch.n_edges, self.E, self.H, self._table_points()
        d_src = batch.edges.reshape(-1)
        tol = -1.0 if self.edge_table_force_fallback else float(self.edge_table_tol)
        reuse = self._frozen and not tape and not training and not self.edge_table_force_fallback
        gate = torch.empty(8, dtype=torch.int32, device=self.device)
        if reuse and self._table_cache is not None and self._table_cache["key"] == (self._wgen, T, E):
            c = self._table_cache
            cover = self._new(4)
            self._ck(lib.ng_edge_table_range(h, st, ne, E, ptr(d_src), ptr(d_eff_c), ptr(pos), None, 0.0, ptr(cover)),
                     "ng_edge_table_range")
            self._ck(lib.ng_edge_table_check(h, st, T, E, None, 0.0, ptr(c["rng"]), ptr(cover), ptr(n_live), 2 * T,
                                             ptr(c["gate"]), ptr(gate)), "ng_edge_table_check")
            return {"e_all": c["e_all"], "rng": c["rng"], "gate": gate, "T": T}
        rng = self._new(4)
        # a table kept over calls covers a quarter more on either side than the call that builds it
        self._ck(lib.ng_edge_table_range(h, st, ne, E, ptr(d_src), ptr(d_eff_c), ptr(pos), None, 0.25 if reuse else 0.0, ptr(rng)),
                 "ng_edge_table_range")
        d_tab, ones = self._new(2 * T), self._new(2 * T)
        perm_tab = torch.empty(2 * T, dtype=torch.int32, device=self.device)
        self._ck(lib.ng_edge_table_points(h, st, T, 1, ptr(rng), ptr(d_tab), ptr(ones), ptr(perm_tab)), "ng_edge_table_points")
        if getattr(self, "_rows_2t", None) is None or int(self._rows_2t_value) != 2 * T:
            self._rows_2t = torch.full((1,), 2 * T, dtype=torch.int32, device=self.device)
            self._rows_2t_value = 2 * T
        e_all = self._new(2 * T, E)
        z_tab = self._new(self.Le - 1, 2 * T, H) if tape else None
        z_layout = int(lib.ng_edge_tape_layout(H, E, self.Le, self.fc_act, 2 * T)) if tape else 0
        W = [P[f"edge_fc/{t}/kernel"] for t in range(self.Le)]
        B = [P[f"edge_fc/{t}/bias"] for t in range(self.Le)]
        # rows 0 .. T-1: the table; rows T .. 2T-1: the midpoints the guard compares at (same launch: the 32 tiles run side by side)
        self._ck(lib.ng_edge_mlp_fwd_live(h, st, 2 * T, H, E, self.Le, self.fc_act, ptr(ones), ptr(d_tab), ptr(perm_tab),
                                          ptr(self._rows_2t), ptr(self.centers), self.gap, ptr_array(W), ptr_array(B),
                                          ptr(e_all), ptr(z_tab)), "ng_edge_mlp_fwd_live")
        self._ck(lib.ng_edge_table_check(h, st, T, E, ptr(e_all), tol, ptr(rng), None, ptr(n_live), 2 * T, None, ptr(gate)),
                 "ng_edge_table_check")
        tb = {"e_all": e_all, "rng": rng, "gate": gate, "T": T, "d_tab": d_tab, "ones": ones, "perm_tab": perm_tab, "z_tab": z_tab,
              "z_layout": z_layout}
        if reuse:
            self._table_cache = {"key": (self._wgen, T, E), "e_all": e_all, "rng": rng, "gate": gate}
        return tb

    def _edge_table_build_sync(self, batch, d_src, d_eff, tape):
        """The table for an edge shape WITHOUT the fused live-edge kernels (layered edge MLP: any hidden size / depth): the same
        passes on the every-slot entry points, and the guard's verdict read on the host — one synchronisation per call instead
        of the device-side gate (nothing on the device can make the layered kernels skip themselves).  None: the guard is up,
        the caller evaluates the MLP per edge as before."""
        lib, h, st = self.lib, self.ctx.handle, self._st()
        P = self.params
        ne, E, H, T = batch.n_edges, self.E, self.H, self._table_points()
        tol = -1.0 if self.edge_table_force_fallback else float(self.edge_table_tol)
        rng = self._new(4)
        self._ck(lib.ng_edge_table_range(h, st, ne, E, ptr(d_src), ptr(d_eff), None, None, 0.0, ptr(rng)), "ng_edge_table_range")
        d_tab, ones = self._new(2 * T), self._new(2 * T)
        self._ck(lib.ng_edge_table_points(h, st, T, 1, ptr(rng), ptr(d_tab), ptr(ones), None), "ng_edge_table_points")
        e_all = self._new(2 * T, E)
        z_tab = self._new(self.Le - 1, 2 * T, H) if tape else None
        z_layout = int(lib.ng_edge_tape_layout(H, E, self.Le, self.fc_act, 2 * T)) if tape else 0
        W = [P[f"edge_fc/{t}/kernel"] for t in range(self.Le)]
        B = [P[f"edge_fc/{t}/bias"] for t in range(self.Le)]
        self._ck(lib.ng_edge_mlp_fwd(h, st, 2 * T, H, E, self.Le, self.fc_act, ptr(ones), ptr(d_tab), ptr(self.centers), self.gap,
                                     ptr_array(W), ptr_array(B), ptr(e_all), ptr(z_tab)), "ng_edge_mlp_fwd")
        gate = torch.empty(8, dtype=torch.int32, device=self.device)
        self._ck(lib.ng_edge_table_check(h, st, T, E, ptr(e_all), tol, ptr(rng), None, None, 2 * T, None, ptr(gate)),
                 "ng_edge_table_check")
        if int(gate[0].item()) != 0:          # the one host round trip of this path
            return None
        return {"e_all": e_all, "rng": rng, "gate": gate, "T": T, "d_tab": d_tab, "ones": ones, "z_tab": z_tab, "z_layout": z_layout}

    def _edge_table_backward_sync(self, tp, de):
        """edge-weight gradients of a call answered by the host-guarded table: scatter de onto the table, the layered backward on
        its 2T rows"""
        lib, h, st = self.lib, self.ctx.handle, self._st()
        P, tb, b = self.params, tp.table_sync, tp.batch
        ne, E, H, T = b.n_edges, self.E, self.H, tb["T"]
        W = [P[f"edge_fc/{t}/kernel"] for t in range(self.Le)]
        dW = [P.g(f"edge_fc/{t}/kernel") for t in range(self.Le)]
        dB = [P.g(f"edge_fc/{t}/bias") for t in range(self.Le)]
        de_tab = self._new(2 * T, E)
        self._ck(lib.ng_edge_table_scatter(h, st, ne, E, T, 2 * T, ptr(b.edges), ptr(tp.d_eff), None, ptr(tb["rng"]), ptr(de),
                                           ptr(de_tab)), "ng_edge_table_scatter")
        self._ck(lib.ng_edge_mlp_bwd_tape(h, st, 2 * T, H, E, self.Le, self.fc_act, ptr(tb["ones"]), ptr(tb["d_tab"]),
                                          ptr(self.centers), self.gap, ptr_array(W), ptr(tb["z_tab"]), ptr(de_tab),
                                          ptr_array(dW), ptr_array(dB), tb["z_layout"]), "ng_edge_mlp_bwd")

    def edge_table_report(self, table=None):
        """(guard up?, interpolation error at the midpoints, largest |e| of the table) of the last taped call's table (or of
        ``table``).  Reads device words: synchronises; for tests and diagnostics."""
        tb = table if table is not None else ((self.tape.table or self.tape.table_sync) if self.tape is not None else None)
        if tb is None:
            return None
        g = tb["gate"].cpu().numpy()
        return bool(g[0]), float(g[4:6].view(np.float32)[0]), float(g[4:6].view(np.float32)[1])

    def _edge_table_backward(self, tp, de):
        """edge-weight gradients of a call that went through the table: the per-edge backward over gate[1] rows (zero unless the
        guard was up) into temporaries, the table's backward over gate[2] rows (zero when it was) into the gradients, and the
        sum of the two — one of which is exactly zero."""
        lib, h, st = self.lib, self.ctx.handle, self._st()
        P, tb, b = self.params, tp.table, tp.batch
        ne, E, H, T = b.n_edges, self.E, self.H, tb["T"]
        perm, pos, d_c, n_live = tp.live
        names = [f"edge_fc/{t}/{k}" for t in range(self.Le) for k in ("kernel", "bias")]
        o0 = min(P.offsets[n] for n in names)
        o1 = max(P.offsets[n] + int(np.prod(P.shapes[n])) for n in names)
        tmp = self._new(o1 - o0)
        tv = lambda n: tmp[P.offsets[n] - o0:P.offsets[n] - o0 + int(np.prod(P.shapes[n]))]
        W = [P[f"edge_fc/{t}/kernel"] for t in range(self.Le)]
        dW = [P.g(f"edge_fc/{t}/kernel") for t in range(self.Le)]
        dB = [P.g(f"edge_fc/{t}/bias") for t in range(self.Le)]
        dWt = [tv(f"edge_fc/{t}/kernel") for t in range(self.Le)]
        dBt = [tv(f"edge_fc/{t}/bias") for t in range(self.Le)]
        self._ck(lib.ng_edge_mlp_bwd_live(h, st, ne, H, E, self.Le, self.fc_act, ptr(d_c), ptr(tp.d_eff), ptr(perm),
                                          ptr(tb["gate"][1:]), ptr(self.centers), self.gap, ptr_array(W), ptr(tp.z_save),
                                          ptr(de), ptr_array(dWt), ptr_array(dBt), tp.z_layout), "ng_edge_mlp_bwd_live")
        de_tab = self._new(2 * T, E)
        self._ck(lib.ng_edge_table_scatter(h, st, ne, E, T, 2 * T, ptr(b.edges), ptr(tp.d_eff), ptr(pos), ptr(tb["rng"]), ptr(de),
                                           ptr(de_tab)), "ng_edge_table_scatter")
        self._ck(lib.ng_edge_mlp_bwd_live(h, st, 2 * T, H, E, self.Le, self.fc_act, ptr(tb["ones"]), ptr(tb["d_tab"]),
                                          ptr(tb["perm_tab"]), ptr(tb["gate"][2:]), ptr(self.centers), self.gap, ptr_array(W),
                                          ptr(tb["z_tab"]), ptr(de_tab), ptr_array(dW), ptr_array(dB), tb["z_layout"]),
                 "ng_edge_mlp_bwd_live")
        blk = P.grad[o0:o1]
        self._ck(lib.ng_add_scaled(h, st, o1 - o0, ptr(blk), ptr(tmp), 1.0, ptr(blk)), "ng_add_scaled")

    # ------------------------------------------------------------------ backward
    def backward(self, dpeaks, on_node_grads=None):
        """fills params.grad (overwrite) from the upstream gradient dpeaks[N] (None: the gradient of the loss
        taken inside ``forward(loss=...)``).
        ``on_node_grads`` is called once every non-edge gradient has been enqueued (the data-parallel
        trainer launches the node-side all-reduce there, overlapping the edge-MLP backward)."""
        tp = self.tape
        if tp is None:
            raise RuntimeError("backward() without forward(training=True)")
        lib, h, st = self.lib, self.ctx.handle, self._st()
        # the seven second-stage sums of the node-side weight gradients (head, FC block, MPLayers, embedding) are queued
        # and run as ONE launch before the node gradients are handed on (ng_defer_reductions: same bits, ~45 us less)
        self._ck(lib.ng_defer_reductions(h, st, 1 if self.defer_reductions else 0), "ng_defer_reductions")
        if self.cache_images:
            self._ck(lib.ng_weights_frozen(h, self._id), "ng_weights_frozen")
        try:
            self._backward(tp, dpeaks, on_node_grads, lib, h, st)
        finally:
            if self.cache_images:
                lib.ng_weights_frozen(h, 0)
            self._ck(lib.ng_defer_reductions(h, st, 0), "ng_defer_reductions")

    def _backward(self, tp, dpeaks, on_node_grads, lib, h, st):
        P = self.params
        b = tp.batch
        N, K, F, E, H = b.N, b.K, self.F, self.E, self.H
        Fh = F // 2
        ne = b.n_edges
        self._ck(lib.ng_ctx_set_graph_span(h, b.max_graph_atoms), "ng_ctx_set_graph_span")
        if dpeaks is None and tp.dg is not None:
            # head + loss + head backward ran as one launch in the forward: only its weight-gradient partials are left
            dg = tp.dg
            self._ck(lib.ng_head_loss_reduce(h, st, ptr(tp.head_partial), tp.head_partial.shape[0], Fh, self.C,
                                             ptr(P.g("out/kernel")), ptr(P.g("out/bias")), ptr(tp.loss)),
                     "ng_head_loss_reduce")
        else:
            if tp.dg is not None:
                raise RuntimeError("backward(dpeaks) after forward(loss=...): the head's backward already ran with the "
                                   "loss gradient; call backward(None)")
            if dpeaks is None:
                dpeaks = tp.dpeaks
            if dpeaks is None:
                raise RuntimeError("backward(None) without forward(loss=...)")
            dpeaks = dpeaks.contiguous()
            dg = self._new(N, Fh)
            self._ck(lib.ng_head_bwd(h, st, N, Fh, self.C, ptr(tp.g), ptr(tp.drop_mask),
                                     ptr(P["out/kernel"]), ptr(b.atoms), ptr(self.peak_std),
                                     ptr(dpeaks), ptr(dg), ptr(P.g("out/kernel")), ptr(P.g("out/bias"))),
                     "ng_head_bwd")
        dx = self._new(N, F)
        Wfc = [P[f"fc/{t}/kernel"] for t in range(self.Lf)]
        ns = int(lib.ng_fc_block_scratch_floats(N, F, self.Lf))
        scratch = self._new(ns) if ns else None
        self._ck(lib.ng_fc_block_bwd(h, st, N, F, self.Lf, self.fc_act, ptr_array(tp.fx), ptr(tp.g),
                                     ptr_array(Wfc), ptr(dg), ptr(dx),
                                     ptr_array([P.g(f"fc/{t}/kernel") for t in range(self.Lf)]),
                                     ptr_array([P.g(f"fc/{t}/bias") for t in range(self.Lf)]),
                                     ptr(scratch)), "ng_fc_block_bwd")
        csc_ptr, csc_edge = b.csc()
        de = self._new(ne, E)
        # incoming-edge records (source atom + edge features in CSC order): shared by all MP layers
        rec = None
        if E <= 3 and not b.is_csr:
            rec = self._new(ne, 4)
            self._ck(lib.ng_mp_edge_records(h, st, N, K, E, ptr(csc_ptr), ptr(csc_edge), ptr(tp.e), ptr(rec)),
                     "ng_mp_edge_records")
        dh = dx
        for l in reversed(range(self.L)):
            dhn = self._new(N, F)
            if b.is_csr:
                self._ck(lib.ng_mp_layer_bwd_csr(h, st, N, ne, F, E, self.mp_act, ptr(tp.h[l]), ptr(b.row_ptr),
                                                 ptr(b.nlist), ptr(b.row_of), ptr(tp.e), ptr(b.inv_degree),
                                                 ptr(P[f"mp/{l}/w"]), ptr(tp.A[l]), ptr(tp.S[l]), ptr(csc_ptr),
                                                 ptr(csc_edge), ptr(dh), ptr(dhn), ptr(de),
                                                 0 if l == self.L - 1 else 1, ptr(P.g(f"mp/{l}/w"))),
                         "ng_mp_layer_bwd_csr")
                dh = dhn
                continue
            self._ck(lib.ng_mp_layer_bwd_rec(h, st, N, K, F, E, self.mp_act, ptr(tp.h[l]), ptr(b.nlist_c),
                                         ptr(tp.e), ptr(b.inv_degree), ptr(P[f"mp/{l}/w"]),
                                         ptr(tp.A[l]), ptr(tp.S[l]), ptr(csc_ptr), ptr(csc_edge),
                                         ptr(dh), ptr(dhn), ptr(de), 0 if l == self.L - 1 else 1,
                                         ptr(P.g(f"mp/{l}/w")), ptr(rec)), "ng_mp_layer_bwd")
            dh = dhn
        self._ck(lib.ng_embed_bwd(h, st, N, self.C, F, ptr(b.atoms), ptr(dh),
                                  ptr(P.g("embed/kernel"))), "ng_embed_bwd")
        self._ck(lib.ng_flush_reductions(h, st), "ng_flush_reductions")
        if on_node_grads is not None:
            on_node_grads()
        W = [P[f"edge_fc/{t}/kernel"] for t in range(self.Le)]
        dW = [P.g(f"edge_fc/{t}/kernel") for t in range(self.Le)]
        dB = [P.g(f"edge_fc/{t}/bias") for t in range(self.Le)]
        if getattr(tp, "table", None) is not None:
            self._edge_table_backward(tp, de)
        elif tp.table_sync is not None:
            self._edge_table_backward_sync(tp, de)
        elif tp.live is not None:
            perm, _, d_c, n_live = tp.live
            self._ck(lib.ng_edge_mlp_bwd_live(h, st, ne, H, E, self.Le, self.fc_act, ptr(d_c), ptr(tp.d_eff), ptr(perm),
                                              ptr(n_live), ptr(self.centers), self.gap, ptr_array(W), ptr(tp.z_save),
                                              ptr(de), ptr_array(dW), ptr_array(dB), tp.z_layout), "ng_edge_mlp_bwd_live")
        else:
            self._ck(lib.ng_edge_mlp_bwd_tape(h, st, ne, H, E, self.Le, self.fc_act, ptr(b.edges), ptr(tp.d_eff),
                                              ptr(self.centers), self.gap, ptr_array(W), ptr(tp.z_save),
                                              ptr(de), ptr_array(dW), ptr_array(dB), tp.z_layout), "ng_edge_mlp_bwd")
        self.tape = None

    # ------------------------------------------------------------------ loss / optimiser
    def loss_l2(self, batch, y, w, peaks):
        """NameLoss with s = 1 (nmrgnn/losses.py:30-39), mean over the batch's graphs.
        Returns (loss[1] device tensor, dloss/dpeaks[N])."""
        loss = self._new(1)
        dpred = self._new(batch.N)
        self._ck(self.lib.ng_loss_l2(self.ctx.handle, self._st(), batch.N, batch.G,
                                     ptr(batch.graph_ptr), ptr(y), ptr(w), ptr(peaks), ptr(loss),
                                     ptr(dpred)), "ng_loss_l2")
        return loss, dpred

    def loss_name(self, batch, y, w, peaks, s=1.0):
        """NameLoss with balance ``s`` (nmrgnn/losses.py:30-39): mean over graphs of
        s*l2 + (1-s)*(1-r).  ``w`` is already the label-filtered weight column.
        Returns (loss[1] device tensor, dloss/dpeaks[N])."""
        if not 0.0 <= float(s) <= 1.0:
            raise ValueError("NameLoss balance s must lie in [0, 1]")
        loss = self._new(1)
        dpred = self._new(batch.N)
        self._ck(self.lib.ng_loss_name(self.ctx.handle, self._st(), batch.N, batch.G,
                                       ptr(batch.graph_ptr), ptr(y), ptr(w), ptr(peaks), float(s),
                                       ptr(loss), ptr(dpred)), "ng_loss_name")
        return loss, dpred

    def adam_step(self, lr=None, grad_scale=1.0, beta1=0.9, beta2=0.999, eps=1e-7):
        if lr is None:
            lr = float(self.hp.get('learning_rate'))
        self.adam_t += 1
        self._wgen += 1
        P = self.params
        if self.cache_images:       # the update is followed by ONE launch that rebuilds this engine's packed weight images
            self._ck(self.lib.ng_weights_frozen(self.ctx.handle, self._id), "ng_weights_frozen")
        try:
            self._ck(self.lib.ng_adam_step(self.ctx.handle, self._st(), P.numel, ptr(P.flat), ptr(P.grad),
                                           ptr(self.adam_m), ptr(self.adam_v), lr, beta1, beta2, eps,
                                           self.adam_t, grad_scale), "ng_adam_step")
        finally:
            if self.cache_images:
                self.lib.ng_weights_frozen(self.ctx.handle, 0)
