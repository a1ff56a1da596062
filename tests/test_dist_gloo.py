"""world_size-2 (gloo, CPU) check of the batch-sharded path: averaged shard gradients == full-batch gradients.

Compute runs on the emulated kernel set (test infrastructure) because there is no GPU here; what is
under test is ``stc_hip.dist`` -- the flat gradient bucket, the equal contiguous sharding and the one
all-reduce -- on the real drop-in modules in ``csr-fixed`` mode with the reference's ComboLoss.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import PKG, REPO


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _build(seed=0):
    import STC_GNN as M
    from stc_hip import CsrGraph
    torch.manual_seed(seed)
    H, W, C, h, K = 4, 5, 3, 4, 2
    graph = CsrGraph.queen_grid(H, W, normalize=True)
    model = M.STCGNN(H * W, C, K, K, 1, h, 2, 2, graph_mode='csr-fixed')
    Gc = torch.softmax(torch.randn(C, C), -1)
    X = (torch.rand(4, 3, H * W, C) < 0.3).float()
    Y = (torch.rand(4, 2, H * W, C) < 0.3).float()
    return model, graph, Gc, X, Y


def _worker(rank, world, port, out_dir):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle import stc_oracle as O
    from oracle.kernel_emul import EmulatedKernels
    from stc_hip import dist as sdist, ops
    ops._kernels = EmulatedKernels()
    r, w, _ = sdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    model, graph, Gc, X, Y = _build()
    bucket = sdist.GradBucket(model.parameters())
    bucket.zero()
    xs, ys = sdist.shard_batch(X, rank, world), sdist.shard_batch(Y, rank, world)
    loss = O.combo_loss(model(X_seq=xs, As=graph, Ac=Gc), ys)
    loss.backward()
    assert bucket.check_views()
    bucket.allreduce_mean()
    torch.save({'flat': bucket.flat.clone(), 'loss': loss.detach()}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_gradients_equal_full_batch(tmp_path):
    world = 2
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    got = [torch.load(tmp_path / f'rank{r}.pt') for r in range(world)]
    assert torch.equal(got[0]['flat'], got[1]['flat'])                    # every rank holds the same averaged bucket

    from oracle import stc_oracle as O
    from oracle.kernel_emul import EmulatedKernels
    from stc_hip import dist as sdist, ops
    old = ops._kernels
    ops._kernels = EmulatedKernels()
    try:
        model, graph, Gc, X, Y = _build()
        bucket = sdist.GradBucket(model.parameters())
        loss = O.combo_loss(model(X_seq=X, As=graph, Ac=Gc), Y)
        loss.backward()
        assert bucket.allreduce_mean() is None                            # no process group: a no-op
        full = bucket.flat.clone()
    finally:
        ops._kernels = old
    denom = float(full.abs().max())
    assert float((got[0]['flat'] - full).abs().max()) / denom < 5e-6
    assert abs(float((got[0]['loss'] + got[1]['loss']) / 2 - loss.detach())) < 1e-6   # equal shards: mean of shard losses
    assert bucket.nbytes == 4 * sum(p.numel() for p in model.parameters())


def _worker_learned(rank, world, port, out_dir):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle import stc_oracle as O
    from oracle.kernel_emul import EmulatedKernels
    from stc_hip import dist as sdist, ops
    ops._kernels = EmulatedKernels()
    sdist.init_from_env(backend='gloo')
    model, As, Ac, X, Y = _build_learned(batch_sharded=True)
    bucket = sdist.GradBucket(model.parameters())
    xs, ys = sdist.shard_batch(X, rank, world), sdist.shard_batch(Y, rank, world)
    Gs, _ = model.mix_graph_pair(xs, As, Ac)                          # graphs of the WHOLE batch on every rank
    bucket.zero()
    O.combo_loss(model(X_seq=xs, As=As, Ac=Ac), ys).backward()
    bucket.allreduce_mean()
    torch.save({'flat': bucket.flat.clone(), 'Gs': Gs.detach()}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def _build_learned(batch_sharded):
    import STC_GNN as M
    torch.manual_seed(11)
    N, C = 12, 3
    model = M.STCGNN(N, C, 2, 2, 1, 4, 1, 2, batch_sharded=batch_sharded)
    As = (torch.rand(N, N) < 0.3).float()
    Ac = torch.rand(C, C)
    X = (torch.rand(4, 3, N, C) < 0.3).float()
    Y = (torch.rand(4, 2, N, C) < 0.3).float()
    return model, As, Ac, X, Y


def test_two_rank_learned_graphs_are_exact_under_batch_sharding(tmp_path):
    """dense-learned mode: MGP_Gen's batch-summed pre-activation is all-reduced (forward and backward), so the
    sharded run reproduces the full-batch graphs and, after the bucket all-reduce, the full-batch gradients."""
    world = 2
    mp.start_processes(_worker_learned, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    got = [torch.load(tmp_path / f'rank{r}.pt') for r in range(world)]
    from oracle import stc_oracle as O
    from oracle.kernel_emul import EmulatedKernels
    from stc_hip import dist as sdist, ops
    old = ops._kernels
    ops._kernels = EmulatedKernels()
    try:
        model, As, Ac, X, Y = _build_learned(batch_sharded=False)
        bucket = sdist.GradBucket(model.parameters())
        Gs, _ = model.mix_graph_pair(X, As, Ac)
        O.combo_loss(model(X_seq=X, As=As, Ac=Ac), Y).backward()
        full = bucket.flat.clone()
    finally:
        ops._kernels = old
    assert float((got[0]['Gs'] - Gs).abs().max()) < 1e-6 and torch.equal(got[0]['Gs'], got[1]['Gs'])
    assert float((got[0]['flat'] - full).abs().max()) / float(full.abs().max()) < 1e-5


def test_shard_batch_and_bucket_guards():
    from stc_hip import dist as sdist
    with pytest.raises(ValueError):
        sdist.shard_batch(torch.zeros(5, 2), 0, 2)
    assert sdist.shard_batch(torch.arange(8).view(8, 1), 1, 4).flatten().tolist() == [2, 3]
    lin = torch.nn.Linear(3, 2)
    bucket = sdist.GradBucket(lin.parameters())
    assert bucket.flat.numel() == 8 and bucket.check_views()
    lin(torch.ones(1, 3)).sum().backward()
    assert float(bucket.flat.abs().sum()) > 0
    bucket.zero()
    assert float(lin.weight.grad.abs().sum()) == 0.0 and bucket.check_views()
    torch.optim.SGD(lin.parameters(), lr=0.1).zero_grad(set_to_none=True)
    assert not bucket.check_views()


def _worker_trainer(rank, world, port, out_dir):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle.kernel_emul import EmulatedKernels
    from stc_hip import data as sdata, dist as sdist, ops
    from stc_hip.trainer import Trainer
    from tests.golden.make_golden import pipeline_inputs
    ops._kernels = EmulatedKernels()
    sdist.init_from_env(backend='gloo')
    data, params = pipeline_inputs()
    params = dict(params, output_dir=out_dir, _allow_cpu_for_tests=True)
    loaders = sdata.get_data_loader(params, data, params['obs_len'], params['pred_len'], params['split_ratio'])
    torch.manual_seed(123 if rank == 0 else 1000 + rank)                  # rank 0 draws the reference's initial parameters; the other
    trainer = Trainer(params, data)                                       # ranks draw different ones and must receive rank 0's (broadcast)
    assert trainer.world == world and trainer.model.mix_graph_pair.batch_sharded
    hist = trainer.train(loaders, verbose=False)
    test = trainer.test(loaders)['test']
    assert trainer.model.mix_graph_pair.batch_sharded                     # restored after the whole-batch evaluation
    torch.save({'hist': hist, 'flat': torch.cat([p.detach().flatten() for p in trainer.model.parameters()]),
                'forecast': test['forecast'], 'bce': test['bce']},
               os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trainer_reproduces_the_reference_epoch_losses(tmp_path):
    """The trainer counterpart (Model_Trainer.py:52-121) batch-sharded over two ranks -- shard_batch on every batch incl.
    the ragged last one (3 samples: 2 + 1), one GradBucket all-reduce per step, learned graphs with the batch-sum
    all-reduce -- follows the REFERENCE's own 2-epoch loss curves (g9), and both ranks end on the same parameters."""
    import numpy as np
    from tests.conftest import load_golden
    world = 2
    mp.start_processes(_worker_trainer, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    got = [torch.load(tmp_path / f'rank{r}.pt', weights_only=False) for r in range(world)]
    g = load_golden('g9_pipeline')
    for r in range(world):
        assert np.allclose(got[r]['hist']['loss']['train'], g['train_curve'].numpy(), rtol=0, atol=3e-5), got[r]['hist']['loss']
        assert np.allclose(got[r]['hist']['loss']['validate'], g['val_curve'].numpy(), rtol=0, atol=3e-5), got[r]['hist']['loss']
    assert torch.equal(got[0]['flat'], got[1]['flat'])
    assert os.path.exists(tmp_path / 'STC-GNN-4.pkl')                     # written once, by rank 0
    # Trainer.test() under two ranks = the one-rank evaluation of the same checkpoint: every rank runs the whole batches, so the
    # learned graphs' batch sum must NOT be all-reduced again (it would come out world times too large before the softmax)
    from oracle.kernel_emul import EmulatedKernels
    from stc_hip import data as sdata, ops
    from stc_hip.trainer import Trainer
    from tests.golden.make_golden import pipeline_inputs
    ops._kernels = EmulatedKernels()
    try:
        data, params = pipeline_inputs()
        params = dict(params, output_dir=str(tmp_path), _allow_cpu_for_tests=True)
        loaders = sdata.get_data_loader(params, data, params['obs_len'], params['pred_len'], params['split_ratio'])
        one = Trainer(params, data).test(loaders)['test']
    finally:
        ops._kernels = None
    for r in range(world):
        assert np.abs(got[r]['forecast'] - one['forecast']).max() < 1e-6 and abs(got[r]['bce'] - one['bce']) < 1e-6


def _worker_short_batch(rank, world, port, out_dir):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle.kernel_emul import EmulatedKernels
    from stc_hip import data as sdata, dist as sdist, ops
    from stc_hip.trainer import Trainer
    from tests.golden.make_golden import pipeline_inputs
    ops._kernels = EmulatedKernels()
    if world > 1:
        sdist.init_from_env(backend='gloo')
    data, params = pipeline_inputs()
    params = dict(params, output_dir=out_dir, _allow_cpu_for_tests=True, num_epochs=1)
    loaders = sdata.get_data_loader(params, data, params['obs_len'], params['pred_len'], params['split_ratio'])
    x, y = next(iter(loaders['train']))
    one = {'train': [(x[:1], y[:1]), (x[1:4], y[1:4])], 'validate': [(x[:1], y[:1])]}      # a 1-sample batch: fewer samples than ranks
    torch.manual_seed(123 if rank == 0 else 77)
    trainer = Trainer(params, data)
    hist = trainer.train(one, verbose=False)
    torch.save({'hist': hist, 'flat': torch.cat([p.detach().flatten() for p in trainer.model.parameters()])}, os.path.join(out_dir, f'w{world}r{rank}.pt'))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_batch_with_fewer_samples_than_ranks_runs_replicated(tmp_path):
    """A batch of ONE sample on two ranks with learned graphs (whose forward all-reduces over the ranks): decided from the batch size
    before any collective, every rank runs the whole small batch -- no rank is left out of a collective, and the step equals the
    one-rank step (losses and final parameters)."""
    mp.start_processes(_worker_short_batch, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True, start_method='spawn')
    mp.start_processes(_worker_short_batch, args=(1, 0, str(tmp_path)), nprocs=1, join=True, start_method='spawn')      # (a process of its own: the worker changes thread counts and the kernel set)
    two = [torch.load(tmp_path / f'w2r{r}.pt', weights_only=False) for r in range(2)]
    one = torch.load(tmp_path / 'w1r0.pt', weights_only=False)
    assert torch.equal(two[0]['flat'], two[1]['flat'])
    assert float((two[0]['flat'] - one['flat']).abs().max()) < 1e-6
    for m in ('train', 'validate'):
        assert abs(two[0]['hist']['loss'][m][0] - one['hist']['loss'][m][0]) < 1e-6


def test_ragged_shards_cover_the_batch():
    from stc_hip import dist as sdist
    for n in (0, 1, 3, 7, 8, 26, 32):
        for world in (1, 2, 3, 8):
            parts = [sdist.shard_batch(torch.arange(n), r, world, ragged=True) for r in range(world)]
            assert torch.equal(torch.cat(parts), torch.arange(n))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


# ---------------------------------------------------------------------------------------------------------------- eight ranks
def _build_eight(seed=5):
    """A graph big enough for the locality analysis (n > 64) handed over in a scrambled node order, and a batch of 11 samples: over eight
    ranks that is 2 + 2 + 2 + 1 + 1 + 1 + 1 + 1 (ragged shards)."""
    import STC_GNN as M
    from stc_hip import CsrGraph
    torch.manual_seed(seed)
    H, W, C, h, K = 9, 8, 3, 4, 2
    graph = CsrGraph.queen_grid(H, W, normalize=True, permute_seed=3)
    model = M.STCGNN(H * W, C, K, K, 1, h, 1, 2, graph_mode='csr-fixed')
    Gc = torch.softmax(torch.randn(C, C), -1)
    X = (torch.rand(11, 2, H * W, C) < 0.3).float()
    Y = (torch.rand(11, 2, H * W, C) < 0.3).float()
    return model, graph, Gc, X, Y


def _worker_eight(rank, world, port, out_dir):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle import stc_oracle as O
    from oracle.kernel_emul import EmulatedKernels
    from stc_hip import dist as sdist, graph as sgraph, ops
    ops._kernels = EmulatedKernels()
    sdist.init_from_env(backend='gloo')
    calls = []
    real = sgraph.CsrGraph.locality_order
    sgraph.CsrGraph.locality_order = lambda self: (calls.append(rank), real(self))[1]
    # (1) csr-fixed, ragged shards: the shard loss weighted by its share of the batch, one bucket all-reduce
    model, graph, Gc, X, Y = _build_eight()
    sdist.broadcast_parameters(model)
    bucket = sdist.GradBucket(model.parameters())
    bucket.zero()
    xs, ys = sdist.shard_batch(X, rank, world, ragged=True), sdist.shard_batch(Y, rank, world, ragged=True)
    loss = O.combo_loss(model(X_seq=xs, As=graph, Ac=Gc), ys) * (len(xs) / len(X) * world)
    loss.backward()
    bucket.allreduce_mean()
    order = graph.with_locality()[1]
    # (2) learned graphs (the reference's own mode): the batch-summed pre-activation all-reduced forward and backward, equal shards of 8
    lm, As, Ac, Xl, Yl = _build_learned(batch_sharded=True)
    Xl, Yl = torch.cat([Xl, Xl.flip(0)]), torch.cat([Yl, Yl.flip(0)])          # 8 samples: one per rank
    sdist.broadcast_parameters(lm)
    lb = sdist.GradBucket(lm.parameters())
    lb.zero()
    xl, yl = sdist.shard_batch(Xl, rank, world), sdist.shard_batch(Yl, rank, world)
    Gs, _ = lm.mix_graph_pair(xl, As, Ac)
    O.combo_loss(lm(X_seq=xl, As=As, Ac=Ac), yl).backward()
    lb.allreduce_mean()
    torch.save({'flat': bucket.flat.clone(), 'order': None if order is None else torch.from_numpy(order), 'rcm_calls': calls, 'shard': len(xs),
                'lflat': lb.flat.clone(), 'Gs': Gs.detach()}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_ragged_shards_learned_graphs_and_one_locality_analysis(tmp_path):
    """The rank count the bench is judged at, on CPU (gloo, one thread per rank, tiny shapes, emulated kernels): ragged contiguous shards of
    an 11-sample batch weighted by their share, the flat bucket's mean, the learned graphs' batch-sum all-reduce -- all equal to the one-process
    full-batch step -- and the host-side node renumbering computed by rank 0 alone and shared (every rank must renumber alike)."""
    world = 8
    mp.start_processes(_worker_eight, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    got = [torch.load(tmp_path / f'rank{r}.pt', weights_only=False) for r in range(world)]
    assert [g_['shard'] for g_ in got] == [2, 2, 2, 1, 1, 1, 1, 1]
    assert got[0]['rcm_calls'] == [0] and all(g_['rcm_calls'] == [] for g_ in got[1:])      # reverse Cuthill-McKee ran once per job
    assert got[0]['order'] is not None and all(torch.equal(g_['order'], got[0]['order']) for g_ in got[1:])
    for key in ('flat', 'lflat', 'Gs'):
        assert all(torch.equal(g_[key], got[0][key]) for g_ in got[1:]), key
    from oracle import stc_oracle as O
    from oracle.kernel_emul import EmulatedKernels
    from stc_hip import dist as sdist, ops
    old = ops._kernels
    ops._kernels = EmulatedKernels()
    try:
        model, graph, Gc, X, Y = _build_eight()
        bucket = sdist.GradBucket(model.parameters())
        O.combo_loss(model(X_seq=X, As=graph, Ac=Gc), Y).backward()
        full = bucket.flat.clone()
        lm, As, Ac, Xl, Yl = _build_learned(batch_sharded=False)
        Xl, Yl = torch.cat([Xl, Xl.flip(0)]), torch.cat([Yl, Yl.flip(0)])
        lb = sdist.GradBucket(lm.parameters())
        Gs, _ = lm.mix_graph_pair(Xl, As, Ac)
        O.combo_loss(lm(X_seq=Xl, As=As, Ac=Ac), Yl).backward()
        lfull = lb.flat.clone()
    finally:
        ops._kernels = old
    assert float((got[0]['flat'] - full).abs().max()) / float(full.abs().max()) < 1e-5
    assert float((got[0]['Gs'] - Gs).abs().max()) < 1e-6
    assert float((got[0]['lflat'] - lfull).abs().max()) / float(lfull.abs().max()) < 1e-5


def _worker_large(rank, world, port, out_dir):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle import stc_oracle as O
    from oracle.kernel_emul import EmulatedKernels
    from stc_hip import dist as sdist, ops
    ops._kernels = EmulatedKernels()
    sdist.init_from_env(backend='gloo')
    model, graph, Gc, X, Y = _build()
    bucket = sdist.GradBucket(model.parameters(), large_bytes=1024)      # every parameter of 256 floats or more reduces by itself
    assert bucket.large and bucket.params
    for step in range(2):                                                # (the second step: zero() dropped the large gradients, nothing accumulated)
        bucket.zero()
        assert all(p.grad is None for p in bucket.large)
        xs, ys = sdist.shard_batch(X, rank, world), sdist.shard_batch(Y, rank, world)
        O.combo_loss(model(X_seq=xs, As=graph, Ac=Gc), ys).backward()
        assert bucket.check_views()
        bucket.allreduce_mean()
    torch.save({n: p.grad.clone() for n, p in model.named_parameters()}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_large_parameters_reduce_outside_the_bucket(tmp_path):
    """GradBucket(large_bytes=...): parameters above the threshold (the reference's MixedFusion matrices: 2 x 400 MB at the SF shape) keep a
    gradient of their own -- no fill, no accumulation into a view -- and are all-reduced one by one; the result equals the one-bucket one."""
    world = 2
    port = _free_port()
    mp.start_processes(_worker_large, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    got = [torch.load(tmp_path / f'rank{r}.pt') for r in range(world)]
    from oracle import stc_oracle as O
    from oracle.kernel_emul import EmulatedKernels
    from stc_hip import dist as sdist, ops
    old = ops._kernels
    ops._kernels = EmulatedKernels()
    try:
        model, graph, Gc, X, Y = _build()
        bucket = sdist.GradBucket(model.parameters())
        assert not bucket.large
        O.combo_loss(model(X_seq=X, As=graph, Ac=Gc), Y).backward()
        want = {n: p.grad.clone() for n, p in model.named_parameters()}
    finally:
        ops._kernels = old
    for name, w in want.items():
        assert torch.equal(got[0][name], got[1][name]), name
        assert float((got[0][name] - w).abs().max()) <= 5e-6 * max(float(w.abs().max()), 1e-3), name


def _worker_locality_failure(rank, world, port, out_dir):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from stc_hip import CsrGraph, dist as sdist
    sdist.init_from_env(backend='gloo')
    graph = CsrGraph.queen_grid(9, 9, normalize=True, permute_seed=3)            # (> 64 nodes: the renumbering analysis runs)
    if rank == 0:
        def broken():
            raise ValueError('no renumbering today')
        graph.locality_order = broken
    try:
        graph.with_locality()
        outcome = 'returned'
    except ValueError as e:                                                       # rank 0: its own exception
        outcome = f'ValueError: {e}'
    except RuntimeError as e:                                                     # the others: told by the status word, not left in the broadcast
        outcome = f'RuntimeError: {e}'
    # a graph only THIS rank uses: the analysis without the collective
    alone = CsrGraph.queen_grid(9, 9, normalize=True, permute_seed=4)
    g2, order = alone.with_locality(collective=False) if rank == 1 else (None, None)
    torch.save({'outcome': outcome, 'alone': None if order is None else int(order.size)}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_a_failed_locality_analysis_on_rank_0_raises_on_every_rank(tmp_path):
    """``CsrGraph.with_locality`` is a collective on the default group (rank 0 computes the node renumbering, the others receive it): an
    exception on rank 0 must reach the other ranks as an exception, not as a hang in the broadcast; ``collective=False`` is the form for a
    graph that only some ranks use."""
    port = _free_port()
    mp.spawn(_worker_locality_failure, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f'rank{r}.pt')) for r in (0, 1))
    assert r0['outcome'] == 'ValueError: no renumbering today'
    assert r1['outcome'].startswith('RuntimeError: CsrGraph: rank 0 failed')
    assert r0['alone'] is None and r1['alone'] == 81
