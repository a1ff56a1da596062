"""Evaluation counterpart of the reference's ``framework/Metrics.py``, without scikit-learn.

    mask_data(x, H, W, mask)                 drop the masked grid cells before evaluation     (Metrics.py:8-33)
    one_step_eval_bi(prob, true, threshold)  the reference's per-step metric dict             (Metrics.py:88-152)
    evaluate_binary(prob, true, threshold)   one dict per horizon step                        (Metrics.py:43-70)

Formulas are the reference's, names included: its "Macro-F1" pools TP/FN/FP over the categories and its
"Micro-F1" averages the per-category scores; thresholds are the per-category historical averages; ROC-AUC is the
rank statistic with ties at half weight and PR-AUC the step-wise average precision, as scikit-learn computes them.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def mask_data(x: np.ndarray, H: int, W: int, mask) -> np.ndarray:
    """(samples, horizon, N, C) -> (samples, horizon, unmasked N, C); also accepts the grid layouts of the reference."""
    assert x.ndim in (4, 5)
    if x.ndim == 4:
        assert x.shape[-2] == H * W
        x = x.reshape(x.shape[0], x.shape[1], H, W, x.shape[-1])
    elif x.shape[-2] == H and x.shape[-1] == W:
        x = x.transpose(0, 1, 3, 4, 2)                       # channel-last
    else:
        assert x.shape[2] == H and x.shape[3] == W
    if mask is None:
        return x.reshape(x.shape[0], x.shape[1], -1, x.shape[-1])
    keep = np.ones((H, W), dtype=bool)
    for (h, w) in mask:
        keep[h, w] = False
    return x[:, :, keep, :]                                  # row-major over the kept cells, as the reference's loop


def _roc_auc(y: np.ndarray, score: np.ndarray) -> float:
    order = np.argsort(score, kind='mergesort')
    s = score[order]
    ranks = np.empty(s.size, dtype=np.float64)
    starts = np.flatnonzero(np.r_[True, s[1:] != s[:-1]])
    ends = np.r_[starts[1:], s.size]
    for a, b in zip(starts, ends):                            # average rank within ties
        ranks[a:b] = 0.5 * (a + b - 1) + 1
    pos = y[order] == 1
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg))


def _average_precision(y: np.ndarray, score: np.ndarray) -> float:
    order = np.argsort(-score, kind='mergesort')
    y, s = y[order], score[order]
    last = np.r_[np.flatnonzero(s[1:] != s[:-1]), s.size - 1]      # last index of each distinct threshold
    tp = np.cumsum(y)[last].astype(np.float64)
    precision = tp / (last + 1)
    recall = tp / tp[-1]
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def one_step_eval_bi(prob: np.ndarray, true: np.ndarray, threshold: Sequence[float], beta: int = 2, precision: int = 4) -> Dict[str, float]:
    """prob, true: (samples, N, C).  The reference's metric dict for one horizon step."""
    assert prob.shape == true.shape and prob.shape[-1] == len(threshold)
    C = prob.shape[-1]
    pred = (prob >= np.asarray(threshold)[None, None, :]).astype(np.int64)
    t = true.astype(np.int64)
    tp = np.array([np.sum((pred[..., c] == 1) & (t[..., c] == 1)) for c in range(C)], dtype=np.float64)
    fn = np.array([np.sum((pred[..., c] == 0) & (t[..., c] == 1)) for c in range(C)], dtype=np.float64)
    fp = np.array([np.sum((pred[..., c] == 1) & (t[..., c] == 0)) for c in range(C)], dtype=np.float64)
    b2 = beta ** 2
    out = {
        'Macro-F1': 2 * tp.sum() / (2 * tp.sum() + fn.sum() + fp.sum()),
        'Micro-F1': np.mean(2 * tp / (2 * tp + fn + fp)),
        f'Macro-F{beta}': (1 + b2) * tp.sum() / ((1 + b2) * tp.sum() + b2 * fn.sum() + fp.sum()),
        f'Micro-F{beta}': np.mean((1 + b2) * tp / ((1 + b2) * tp + b2 * fn + fp)),
    }
    p, y, yb = prob.reshape(-1).astype(np.float64), t.reshape(-1), pred.reshape(-1)
    out['Recall'] = float(np.sum((yb == 1) & (y == 1)) / np.sum(y == 1))
    out['ROC-AUC'] = _roc_auc(y, p)
    out['PR-AUC'] = _average_precision(y, p)
    eps = np.finfo(prob.dtype if np.issubdtype(prob.dtype, np.floating) else np.float64).eps
    pc = np.clip(prob.reshape(-1), eps, 1 - eps).astype(np.float64)
    out['BCE'] = float(-np.mean(y * np.log(pc) + (1 - y) * np.log(1 - pc)))
    out['MAE'] = float(np.mean(np.abs(y - p)))
    return {k: round(float(v), precision) for k, v in out.items()}


def evaluate_binary(prob: np.ndarray, true: np.ndarray, threshold: Sequence[float], beta: int = 2) -> List[Dict[str, float]]:
    """prob, true: (samples, horizon, N, C) -> the reference's per-step metrics, one dict per horizon step."""
    assert prob.shape == true.shape
    return [one_step_eval_bi(prob[:, s], true[:, s], threshold, beta) for s in range(prob.shape[1])]
