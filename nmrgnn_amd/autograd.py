"""The outer autograd shell (SURVEY §7, §8b "who calls it"): ``loss = f(model(g)); loss.backward()`` fills the
gradient of the model's parameters, as the reference trains by autodiff through ``model(x)``
(nmrgnn/main.py:74-80, keras ``fit`` over nmrgnn/model.py:245-274).

torch.autograd differentiates the USER's loss; the model itself is one node of the autograd graph whose
forward / backward are ``Engine.forward`` / ``Engine.backward`` — the hand-written HIP kernels and their
hand-derived reverse pass.  The trainable state is ONE leaf: the engine's flat fp32 parameter buffer
(``params.ParamStore.flat``), wrapped without a copy; per-tensor parameters are views into it
(``GNNModel.named_parameter_views``), so ``torch.optim.Adam([flat], eps=1e-7)`` is the Keras optimiser
(nmrgnn/model.py:44-45) elementwise and ``ng_adam_step`` remains the fused equivalent.
"""
from __future__ import annotations

import torch


class GNNModelFunction(torch.autograd.Function):
    """peaks[N] = GNNModel.call((atoms, nlist, edges, inv_degree)) as an autograd node over the flat parameter leaf.

    Inputs of the graph tuple receive no gradient (the reference's inputs are constants of the traced graph:
    nlist / mask / inv_degree have none).  ``backward`` may run once per forward (the tape is
    released, like ``retain_graph=False``)."""

    @staticmethod
    def forward(ctx, flat, engine, batch, training, seed, noise, dropout_mask):
        if flat.data_ptr() != engine.params.flat.data_ptr():
            raise ValueError("GNNModelFunction: the parameter leaf is not this engine's flat buffer")
        # torch-side optimisers write into the buffer behind the library's back
        engine.weights_changed()
        peaks = engine.forward(batch, training=training, noise=noise, dropout_mask=dropout_mask, seed=seed,
                               keep_tape=True)
        ctx.engine = engine
        ctx.tape = engine.tape
        engine.tape = None            # the tape belongs to THIS node: several forwards may be alive at once
        return peaks

    @staticmethod
    def backward(ctx, dpeaks):
        engine, tape = ctx.engine, ctx.tape
        if tape is None:
            raise RuntimeError("GNNModelFunction.backward: the tape was already consumed (backward twice)")
        ctx.tape = None
        engine.tape = tape
        engine.backward(dpeaks.contiguous().to(torch.float32))
        # Engine.backward OVERWRITES params.grad; autograd accumulates into leaf.grad, so hand it a copy it may keep
        return engine.params.grad.clone(), None, None, None, None, None, None


def model_forward(engine, flat_leaf, batch, training=False, seed=0, noise=None, dropout_mask=None):
    return GNNModelFunction.apply(flat_leaf, engine, batch, bool(training), int(seed), noise, dropout_mask)


class KerasAdam(torch.optim.Optimizer):
    """``torch.optim``-shaped front of the fused ``ng_adam_step`` kernel: Keras Adam as the reference compiles it
    (nmrgnn/model.py:44-45 — lr 1e-4, beta 0.9 / 0.999, epsilon 1e-7, ``w -= lr_t * m / (sqrt(v) + eps)`` with
    ``lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)``).  ``torch.optim.Adam(eps=1e-7)`` is NOT the same update: torch adds its
    epsilon to ``sqrt(v_hat)``, i.e. an effective ``eps * sqrt(1 - b2^t)`` in the Keras form — 30x smaller at t = 1 — so
    parameters whose gradients are of the order of epsilon move differently.  Use this class for reference semantics.

    ``params`` must be ``model.parameters()`` (the single flat leaf of the engine)."""

    def __init__(self, model, lr=None, betas=(0.9, 0.999), eps=1e-7):
        (leaf,) = model.parameters()
        if lr is None:
            lr = float(model.hypers.get('learning_rate'))
        super().__init__([leaf], dict(lr=lr, betas=betas, eps=eps))
        self._engine = model.engine

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng = self._engine
        g = self.param_groups[0]
        (leaf,) = g["params"]
        if leaf.grad is None:
            return loss
        grad = leaf.grad.contiguous()
        from ._lib import ptr
        eng.adam_t += 1
        P = eng.params
        eng._ck(eng.lib.ng_adam_step(eng.ctx.handle, eng._st(), P.numel, ptr(P.flat), ptr(grad), ptr(eng.adam_m),
                                     ptr(eng.adam_v), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                     float(g["eps"]), eng.adam_t, 1.0), "ng_adam_step")
        return loss
