"""``NameLoss`` (nmrgnn/losses.py:17-39).  The s = 1 case (weighted L2 over the atoms whose name id
is in ``label_idx``) runs on the GPU through ng_loss_l2 inside the trainer; this class mirrors the
reference's callable for single graphs on the host (s < 1 adds the (1 - r) correlation term)."""
from __future__ import annotations

import numpy as np


def corr_coeff(x, y, w=None):
    """nmrgnn/losses.py:4-15.  Deliberate deviation: sum(w) == 0 gives r = 0 (finite loss) where the reference's
    0/0 moments give NaN; ng_loss_name does the same."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    w = np.ones_like(x) if w is None else np.asarray(w, np.float64)
    m = w.sum()
    if m == 0:
        return 0.0
    xm, ym = (w * x).sum() / m, (w * y).sum() / m
    xm2, ym2 = (w * x ** 2).sum() / m, (w * y ** 2).sum() / m
    cov = (w * (x - xm) * (y - ym)).sum()
    den = m * np.sqrt(np.clip((xm2 - xm ** 2) * (ym2 - ym ** 2), 0, 1e32))
    return float(cov / den) if den != 0 else 0.0


class NameLoss:
    def __init__(self, label_idx=None, s=1., name='name-loss', reduction='none'):
        self.label_idx = label_idx
        self.s = s
        self.name = name

    def get_config(self):
        return {'label_idx': self.label_idx, 's': self.s}

    def weights(self, y_true):
        y_true = np.asarray(y_true)
        w = y_true[:, -1].astype(np.float64)
        if self.label_idx is not None:
            ln = np.asarray(self.label_idx, np.int32)
            w = w * np.any(y_true[:, 1].astype(np.int32)[:, None] == ln[None, :], axis=-1)
        return w

    def __call__(self, y_true, y_pred, sample_weight=None):
        y_true = np.asarray(y_true, np.float64)
        x = np.asarray(y_pred, np.float64)
        w = self.weights(y_true)
        y = y_true[:, 0]
        sw = w.sum()
        l2 = float((w * (y - x) ** 2).sum() / sw) if sw != 0 else 0.0
        if self.s == 1.0:
            return l2
        return l2 * self.s + (1 - self.s) * (1 - corr_coeff(x, y, w))

    call = __call__
