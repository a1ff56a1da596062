"""Evaluation counterpart of the reference's ``framework/Metrics.py`` on the device, without scikit-learn.

    mask_data(x, H, W, mask)                 drop the masked grid cells before evaluation     (Metrics.py:8-33)
    one_step_eval_bi(prob, true, threshold)  the reference's per-step metric dict             (Metrics.py:88-152)
    evaluate_binary(prob, true, threshold)   one dict per horizon step                        (Metrics.py:43-70)
    append_metrics_csv(path, params, ...)    the reference's ``<model>_eval-bi-metrics.csv`` log (Metrics.py:54-59, 72-85)

Inputs may be torch tensors on any device (the trainer's predictions stay on the GPU) or numpy arrays; the work is
torch ops on the input's device -- confusion counts, two sorts for the AUCs -- in float64, and only the handful of
scalars of the result come back to the host.  Formulas are the reference's, names included: its "Macro-F1" pools
TP/FN/FP over the categories and its "Micro-F1" averages the per-category scores; thresholds are the per-category
historical averages; ROC-AUC is the rank statistic with ties at half weight and PR-AUC the step-wise average precision,
as scikit-learn computes them.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch


def _t(x) -> torch.Tensor:
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))


def mask_data(x, H: int, W: int, mask):
    """(samples, horizon, N, C) -> (samples, horizon, unmasked N, C); also accepts the grid layouts of the reference.
    Returns the input's kind (tensor on the same device, or numpy array)."""
    as_numpy = not isinstance(x, torch.Tensor)
    x = _t(x)
    assert x.dim() in (4, 5)
    if x.dim() == 4:
        assert x.shape[-2] == H * W
        x = x.reshape(x.shape[0], x.shape[1], H, W, x.shape[-1])
    elif x.shape[-2] == H and x.shape[-1] == W:
        x = x.permute(0, 1, 3, 4, 2)                         # channel-last
    else:
        assert x.shape[2] == H and x.shape[3] == W
    if mask is None:
        out = x.reshape(x.shape[0], x.shape[1], -1, x.shape[-1])
    else:
        keep = torch.ones(H, W, dtype=torch.bool)
        for (h, w) in mask:
            keep[h, w] = False
        out = x[:, :, keep.to(x.device), :]                  # row-major over the kept cells, as the reference's loop
    return out.cpu().numpy() if as_numpy else out


def _roc_auc(y: torch.Tensor, score: torch.Tensor) -> float:
    order = torch.argsort(score, stable=True)
    s = score[order]
    _, inv, counts = torch.unique_consecutive(s, return_inverse=True, return_counts=True)
    ends = counts.cumsum(0).double()
    avg_rank = (2 * ends - counts.double() + 1) / 2              # mean of the 1-based ranks start+1 .. end within a tie
    ranks = avg_rank[inv]
    pos = y[order] == 1
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg))


def _average_precision(y: torch.Tensor, score: torch.Tensor) -> float:
    order = torch.argsort(score, descending=True, stable=True)
    y, s = y[order], score[order]
    _, counts = torch.unique_consecutive(s, return_counts=True)
    last = counts.cumsum(0) - 1                                       # last index of each distinct threshold
    tp = y.double().cumsum(0)[last]
    precision = tp / (last + 1).double()
    recall = tp / tp[-1]
    prev = torch.cat([recall.new_zeros(1), recall[:-1]])
    return float(((recall - prev) * precision).sum())


def one_step_eval_bi(prob, true, threshold: Sequence[float], beta: int = 2, precision: int = 4) -> Dict[str, float]:
    """prob, true: (samples, N, C).  The reference's metric dict for one horizon step."""
    prob, true = _t(prob), _t(true).to(_t(prob).device)
    assert prob.shape == true.shape and prob.shape[-1] == len(threshold)
    thr = torch.as_tensor(np.asarray(threshold, dtype=np.float64), device=prob.device).to(prob.dtype)
    pred = prob >= thr
    t = true.round().bool()
    dims = tuple(range(prob.dim() - 1))
    tp = (pred & t).sum(dims).double()
    fn = (~pred & t).sum(dims).double()
    fp = (pred & ~t).sum(dims).double()
    b2 = beta ** 2
    out = {
        'Macro-F1': 2 * tp.sum() / (2 * tp.sum() + fn.sum() + fp.sum()),
        'Micro-F1': (2 * tp / (2 * tp + fn + fp)).mean(),
        f'Macro-F{beta}': (1 + b2) * tp.sum() / ((1 + b2) * tp.sum() + b2 * fn.sum() + fp.sum()),
        f'Micro-F{beta}': ((1 + b2) * tp / ((1 + b2) * tp + b2 * fn + fp)).mean(),
    }
    p, y = prob.reshape(-1), t.reshape(-1).long()
    out['Recall'] = tp.sum() / (tp.sum() + fn.sum())
    out['ROC-AUC'] = _roc_auc(y, p)
    out['PR-AUC'] = _average_precision(y, p)
    eps = torch.finfo(prob.dtype if prob.dtype.is_floating_point else torch.float64).eps
    pc = p.clamp(eps, 1 - eps).double()
    yd = y.double()
    out['BCE'] = -(yd * pc.log() + (1 - yd) * (1 - pc).log()).mean()
    out['MAE'] = (yd - p.double()).abs().mean()
    return {k: round(float(v), precision) for k, v in out.items()}


def evaluate_binary(prob, true, threshold: Sequence[float], beta: int = 2) -> List[Dict[str, float]]:
    """prob, true: (samples, horizon, N, C) -> the reference's per-step metrics, one dict per horizon step."""
    prob, true = _t(prob), _t(true)
    assert prob.shape == true.shape
    return [one_step_eval_bi(prob[:, s], true[:, s], threshold, beta) for s in range(prob.shape[1])]


def append_metrics_csv(path: str, params: dict, mode: str, multistep_metrics: List[Dict[str, float]]) -> None:
    """Append one evaluation to the reference's CSV log, same layout (``Metrics.py:54-59, 72-85``): a start marker, the full
    parameter dump on one line, a header row (' ', metric names), one row per horizon step, an end marker and a blank line."""
    import time
    with open(path, 'a') as cf:
        cf.write(f'*****, Evaluation starts, {mode}, {time.ctime()}, ***** \n')
        for key in params.keys():
            cf.write(f'{key}: {params[key]},')
        cf.write('\n')
        names = list(multistep_metrics[0].keys()) if multistep_metrics else []
        cf.write(','.join([' '] + names) + '\n')
        for step, m in enumerate(multistep_metrics):
            cf.write(','.join([f'Step {step}'] + [str(v) for v in m.values()]) + '\n')
        cf.write(f'*****, Evaluation ends, {mode}, {time.ctime()}, ***** \n \n')
