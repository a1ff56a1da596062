"""Data pipeline counterpart of the reference's ``framework/Data_Container.py``.

Same file schema, same sliding windows, same contiguous split -- but the whole window set is a strided VIEW of
one device-resident tensor and a batch is one slice of it, instead of a Python list of per-window copies
indexed item by item through ``torch.utils.data.DataLoader`` (``Data_Container.py:44-67, 102, 107-112``).

    load_incidents(path)                    npz -> dict(inc, mask, HA, s_adj, c_cor)          (:10-21)
    sliding_windows(data, obs, pred)        x[i] = data[i:i+obs], y[i] = data[i+obs:i+obs+pred]   (:107-112)
    split_lengths(n, ratio)                 validate/test by floor, train = the rest          (:76-82)
    DeviceBatches / get_data_loader(...)    iterable of (x_seq, y_true) batches per mode, shuffle=False (:84-105)
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np
import torch

MODES = ('train', 'validate', 'test')


def load_incidents(path: str) -> dict:
    """The reference's dataset dict from an ``*-incidents-*h.npz`` file (``Data_Container.py:10-21``)."""
    with np.load(path) as z:
        return dict(inc=z['incident'], mask=[tuple(a) for a in z['mask']], HA=z['threshold'],
                    s_adj=z['s_adj'], c_cor=z['c_cor'])


def synthetic_incidents(H: int, W: int, C: int, T: int, rate: float = 0.1635, seed: int = 0, sparse_graph: bool = False) -> dict:
    """A dataset dict of the reference's schema without a city file: Bernoulli(``rate``) incidents (T, H, W, C)
    (0.1635 = the SF file's mean incidence), the H x W 8-neighbour adjacency (``s_adj`` of the SF file for 10 x 10) -- dense 0/1 as
    in the file, or with ``sparse_graph`` a row-normalised ``CsrGraph`` for ``csr-fixed`` training at sizes where N x N does not
    exist -- a seeded category graph, per-category historical averages as thresholds, no masked cells."""
    from .graph import CsrGraph
    g = torch.Generator().manual_seed(seed)
    inc = (torch.rand(T, H, W, C, generator=g) < rate).to(torch.int32).numpy()
    c_cor = torch.softmax(torch.randn(C, C, generator=torch.Generator().manual_seed(7)), -1).double().numpy()
    graph = CsrGraph.queen_grid(H, W, normalize=sparse_graph)
    s_adj = graph if sparse_graph else graph.to_dense().double().numpy()
    return dict(inc=inc, mask=[], HA=inc.reshape(T, -1, C).mean((0, 1)), s_adj=s_adj, c_cor=c_cor)


def sliding_windows(data: torch.Tensor, obs_len: int, pred_len: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(x, y) views of ``data`` (T, N, C): window w observes data[w:w+obs] and predicts data[w+obs:w+obs+pred].

    The reference enumerates i in [obs, T-pred) with x = data[i-obs:i], y = data[i:i+pred]
    (``Data_Container.py:107-112``): T - obs - pred windows, the last possible one is not used.  No copy is made.
    """
    T = data.shape[0]
    n = T - obs_len - pred_len
    if n <= 0:
        raise ValueError(f'series of {T} steps is too short for obs={obs_len} + pred={pred_len}')
    x = data.unfold(0, obs_len, 1).movedim(-1, 1)[:n]                       # (n, obs, N, C)
    y = data[obs_len:].unfold(0, pred_len, 1).movedim(-1, 1)[:n]            # (n, pred, N, C)
    return x, y


def split_lengths(n_windows: int, ratio: Sequence[int]) -> Dict[str, int]:
    """Contiguous train : validate : test lengths (``Data_Container.py:76-82``)."""
    total = sum(ratio)
    out = {'validate': int(ratio[1] / total * n_windows), 'test': int(ratio[2] / total * n_windows)}
    out['train'] = n_windows - out['validate'] - out['test']
    return out


class DeviceBatches:
    """Contiguous batches [start, start+length) of a window set, in order (the reference's ``shuffle=False``)."""

    def __init__(self, x: torch.Tensor, y: torch.Tensor, start: int, length: int, batch_size: int):
        self.x, self.y, self.start, self.length, self.batch_size = x, y, start, length, batch_size

    def __len__(self):
        return (self.length + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for lo in range(self.start, self.start + self.length, self.batch_size):
            hi = min(lo + self.batch_size, self.start + self.length)
            yield self.x[lo:hi].contiguous(), self.y[lo:hi].contiguous()


def get_data_loader(params: dict, data: dict, obs_len: int, pred_len: int, split_ratio: Sequence[int]) -> Dict[str, DeviceBatches]:
    """``DataGenerator.get_data_loader`` (``Data_Container.py:84-105``): (T,H,W,C) -> (T,N,C) float32 on ``params['device']``."""
    inc = np.asarray(data['inc'])
    series = torch.from_numpy(inc.reshape(inc.shape[0], params['H'] * params['W'], params['C'])).float()
    series = series.to(params['device'])
    x, y = sliding_windows(series, obs_len, pred_len)
    lens = split_lengths(x.shape[0], split_ratio)
    starts = {'train': 0, 'validate': lens['train'], 'test': lens['train'] + lens['validate']}
    return {m: DeviceBatches(x, y, starts[m], lens[m], params['batch_size']) for m in MODES}
