"""Batch-sharded training across the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference has no distributed code.  The hot path shards naturally over the
batch (= sliding-window sample) dimension and only there: every op keeps ``b`` independent,
time is a true recurrence, the graph is replicated.  So the data path needs no
collective at all; the single exchange per iteration is the sum of the parameter gradients.

``GradBucket`` keeps every parameter's ``.grad`` as a view into ONE flat fp32 buffer, so the
exchange is a single in-place ``all_reduce`` (``backend='nccl'`` is RCCL on ROCm) with no
pack/unpack copies.  In ``csr-fixed`` mode the bucket is 22 033 floats (88 KB) at h=16, K=2,
2 layers: latency-bound, one call.  With equal shards and the reference's ComboLoss (mean BCE +
per-sample Dice averaged over the batch, Model_Trainer.py:14-23) the averaged shard gradients
equal the single-process full-batch gradients exactly (tests/test_dist_gloo.py).

``dense-learned`` mode couples the batch: MGP_Gen sums its pre-activation over batch and time before the
softmax (STC_GNN.py:231), so the graphs depend on the whole mini-batch.  ``allreduce_sum`` makes that
exact under sharding: the (N x N and C x C) partial sums are all-reduced in forward and their gradient in
backward -- ``STCGNN(..., batch_sharded=True)``; two small collectives per step on top of the bucket.
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> tuple:
    """[lo, hi) of contiguous shard ``rank`` of ``world`` over ``n`` samples; the first ``n % world`` shards hold one more."""
    per, extra = divmod(n, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def shard_batch(t: torch.Tensor, rank: int, world: int, ragged: bool = False) -> torch.Tensor:
    """Contiguous shard ``rank`` of ``world`` along dim 0 (the sample dimension).

    Equal shards by default (then the plain mean of the shard gradients IS the full-batch gradient).  ``ragged=True``
    accepts any batch size -- the last batch of an epoch -- with shard sizes differing by at most one (possibly empty);
    the caller then scales the shard loss by ``len(shard) / n * world`` before ``backward()`` (see ``GradBucket.allreduce_mean``)."""
    n = t.shape[0]
    if n % world and not ragged:
        raise ValueError(f'global batch {n} is not divisible by world size {world}: shards must be equal '
                         'for the averaged gradients to equal the full-batch gradients')
    lo, hi = shard_bounds(n, rank, world)
    return t[lo:hi]


class _Works:
    """The handles of several asynchronous collectives as one."""

    def __init__(self, works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()


class GradBucket:
    """All gradients of ``params`` in one flat buffer; ``allreduce_mean()`` is the only collective.

    A parameter of ``large_bytes`` or more stays OUT of the buffer (the reference's ``MixedFusion`` holds two (n^2, n^2) matrices -- 2 x 400 MB
    at the SF shape, 99.99 % of the model): in the buffer its gradient costs a fill of its bytes per step plus autograd's read-add-write into
    the view (3 x its bytes) where a fresh ``.grad`` is just the tensor the backward kernel wrote -- 0.6 ms of the 7.6 ms learned-graph SF
    step.  ``zero()`` drops such a gradient (``None``), ``allreduce_mean()`` reduces it by itself: a message of that size needs no bucketing.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], large_bytes: int = 64 << 20):
        every = [p for p in params if p.requires_grad]
        if not every:
            raise ValueError('no trainable parameters')
        dev = {p.device for p in every}
        dt = {p.dtype for p in every}
        if len(dev) != 1 or len(dt) != 1:
            raise ValueError(f'parameters must share one device and dtype, got {dev} / {dt}')
        self.large: List[torch.nn.Parameter] = [p for p in every if p.numel() * p.element_size() >= large_bytes]
        self.params: List[torch.nn.Parameter] = [p for p in every if p.numel() * p.element_size() < large_bytes]
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=every[0].dtype, device=every[0].device)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)      # autograd accumulates into the view in place
            off += n
        for p in self.large:
            p.grad = None

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    @property
    def large_nbytes(self) -> int:
        return sum(p.numel() * p.element_size() for p in self.large)

    def zero(self):
        """Replaces ``optimizer.zero_grad()``: keeps the views, zeroes the storage in one fill (large parameters: gradient dropped)."""
        self.flat.zero_()
        for p in self.large:
            p.grad = None

    def check_views(self):
        """True while every bucketed ``.grad`` still aliases the bucket (``zero_grad(set_to_none=True)`` breaks it)."""
        base = self.flat.untyped_storage().data_ptr()
        return all(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in self.params)

    def allreduce_mean(self, group=None, async_op: bool = False):
        """Sum the gradients over ranks and divide by the world size (gradient of the global-batch mean loss): the bucket in one collective,
        every large parameter in its own.

        Unequal shards (``shard_batch(..., ragged=True)``): scale the shard loss by ``len(shard) / n * world`` before
        ``backward()``; the mean over ranks is then the share-weighted sum = the full-batch gradient (ComboLoss is a mean of
        per-sample and per-element terms, Model_Trainer.py:14-23), also through ``allreduce_sum`` (learned graphs)."""
        if not (dist.is_available() and dist.is_initialized()):
            return None
        world = dist.get_world_size(group)
        if world == 1:
            return None
        if not self.check_views():
            raise RuntimeError('a parameter .grad no longer aliases the bucket: use bucket.zero(), not '
                               'optimizer.zero_grad(set_to_none=True)')
        self.flat.div_(world)                                # pre-scale: the sum then is the mean
        works = [dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)] if self.flat.numel() else []
        for p in self.large:
            if p.grad is None:                               # (every rank ran the same graph: the parameter was unused everywhere)
                continue
            p.grad.div_(world)
            works.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group, async_op=async_op))
        return _Works(works) if async_op else None


class _AllReduceSum(torch.autograd.Function):
    """y = sum over ranks of x, differentiable: the gradient of a replicated consumer w.r.t. one rank's summand is
    the sum of every rank's local gradient (the gradient bucket then averages parameter gradients over ranks, so
    together they yield d(mean over ranks of the shard losses)/d(theta))."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        y = x.contiguous().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


def allreduce_sum(x: torch.Tensor, group=None) -> torch.Tensor:
    """Differentiable cross-rank sum; identity when no process group is initialised or it has one rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x
    return _AllReduceSum.apply(x, group)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Every rank starts from rank ``src``'s parameters and buffers.  Each rank builds the model from its own RNG state; the gradient
    bucket only averages GRADIENTS, so without this the replicas would apply the same update to different points and drift apart."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


def init_from_env(backend: str = None) -> tuple:
    """(rank, world, local_rank) from torchrun's environment; initialises the process group whenever the process was
    started by a launcher (``RANK`` is set) -- also for a single rank, so that a one-GPU ``torchrun`` run goes through the
    same RCCL communicator set-up, collectives and tear-down as an eight-GPU one.

    Test hooks (a 1-GPU box cannot run RCCL between two ranks): ``STC_DIST_BACKEND=gloo`` forces the backend and
    ``STC_DIST_ONE_DEVICE=1`` maps every rank onto device 0, so the N > 1 code path can be exercised on one GPU."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('STC_DIST_ONE_DEVICE') == '1':
        local = 0
    if (world > 1 or 'RANK' in os.environ) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL shares device buffers between the ranks of a node through IPC handles; this driver stack supports the dmabuf form only
        # (without it: hipIpcGetMemHandle "invalid argument" at communicator set-up).  Set before the communicator exists, for every
        # way a rank can be started (bench.py's own launcher, the driver's torchrun form, tools/train.py).
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = backend or os.environ.get('STC_DIST_BACKEND')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local
