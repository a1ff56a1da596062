"""SURVEY 8(f1, f2): data-pipeline and trainer counterparts against the reference's Data_Container / Model_Trainer.

Goldens (tests/golden/g9_pipeline.npz) come from the reference run on a small synthetic incident series; the
trainer runs here on the emulated kernel set (CPU) -- the GPU run of the same loop is in test_module_parity's
Adam-trajectory case and tools/train.py.
"""
import os
import re

import numpy as np
import pytest
import torch

from oracle.kernel_emul import EmulatedKernels
from stc_hip import data as sdata, ops
from stc_hip.trainer import Trainer
from tests.conftest import REPO, load_golden, sub_dict
from tests.golden.make_golden import pipeline_inputs

SF = '/root/reference/data/SF-incidents-4h.npz'


@pytest.mark.parametrize('device', ['cpu', pytest.param('cuda:0', marks=pytest.mark.gpu)])
def test_windows_split_and_batches_match_the_reference(device):
    """Windows, the contiguous split and the batches against what the reference's DataGenerator / DataLoader produced (golden g9;
    Data_Container.py:84-112) -- bit for bit, with the series resident on the CPU and on the GPU (``params['device']``, as Main.py:11-12 passes it)."""
    g = load_golden('g9_pipeline')
    data, params = pipeline_inputs()
    params = dict(params, device=device)
    loaders = sdata.get_data_loader(params, data, params['obs_len'], params['pred_len'], params['split_ratio'])
    for mode in sdata.MODES:
        ld = loaders[mode]
        assert ld.length == int(g[f'{mode}_len']) and len(ld) == int(g[f'{mode}_batches'])
        batches = list(ld)
        assert all(t.device == torch.device(device) for b in batches for t in b[:2])
        assert torch.equal(batches[0][0].cpu(), g[f'{mode}_x0']) and torch.equal(batches[0][1].cpu(), g[f'{mode}_y0'])
        assert torch.equal(batches[-1][0].cpu(), g[f'{mode}_xlast']) and torch.equal(batches[-1][1].cpu(), g[f'{mode}_ylast'])
        assert batches[0][0].dtype == torch.float32 and batches[0][0].is_contiguous()


def test_sliding_windows_are_views_and_edge_cases():
    series = torch.arange(10 * 2 * 1, dtype=torch.float32).view(10, 2, 1)
    x, y = sdata.sliding_windows(series, 3, 2)
    assert x.shape == (5, 3, 2, 1) and y.shape == (5, 2, 2, 1)            # T - obs - pred windows
    assert x.untyped_storage().data_ptr() == series.untyped_storage().data_ptr()   # no copy
    assert torch.equal(x[0], series[0:3]) and torch.equal(y[0], series[3:5])
    assert torch.equal(x[4], series[4:7]) and torch.equal(y[4], series[7:9])
    with pytest.raises(ValueError):
        sdata.sliding_windows(series, 8, 2)
    assert sdata.split_lengths(5112, [6, 1, 1]) == {'validate': 639, 'test': 639, 'train': 3834}
    assert sdata.split_lengths(7, [1, 3, 3]) == {'validate': 3, 'test': 3, 'train': 1}


@pytest.mark.skipif(not os.path.exists(SF), reason='SF data file not present')
def test_sf_file_counts():
    data = sdata.load_incidents(SF)
    assert data['inc'].shape == (5124, 10, 10, 5) and data['s_adj'].shape == (100, 100) and data['c_cor'].shape == (5, 5)
    params = dict(device='cpu', H=10, W=10, C=5, batch_size=32)
    loaders = sdata.get_data_loader(params, data, 9, 3, [6, 1, 1])
    assert [loaders[m].length for m in sdata.MODES] == [3834, 639, 639]
    x0, y0 = next(iter(loaders['train']))
    inc = torch.from_numpy(data['inc'].reshape(5124, 100, 5)).float()
    assert torch.equal(x0[0], inc[0:9]) and torch.equal(y0[0], inc[9:12]) and len(loaders['train']) == 120


def test_trainer_reproduces_the_reference_epoch_losses(tmp_path, monkeypatch):
    """Same seed -> same initial parameters (same creation order) -> the reference's 2-epoch loss curves."""
    monkeypatch.setattr(ops, '_kernels', EmulatedKernels())
    g = load_golden('g9_pipeline')
    data, params = pipeline_inputs()
    params = dict(params, output_dir=str(tmp_path), _allow_cpu_for_tests=True)
    loaders = sdata.get_data_loader(params, data, params['obs_len'], params['pred_len'], params['split_ratio'])
    torch.manual_seed(123)
    trainer = Trainer(params, data)
    for k, v in sub_dict(g, 'sd0/').items():                         # identical initialisation, key for key
        assert torch.equal(trainer.model.state_dict()[k], v), k
    hist = trainer.train(loaders, verbose=False)
    assert np.allclose(hist['loss']['train'], g['train_curve'].numpy(), rtol=0, atol=2e-5)
    assert np.allclose(hist['loss']['validate'], g['val_curve'].numpy(), rtol=0, atol=2e-5)
    ck = torch.load(trainer.checkpoint_path)
    assert sorted(ck.keys()) == ['epoch', 'state_dict', 'train_loss', 'val_loss'] and ck['epoch'] == int(g['ckpt_epoch'])
    assert os.path.basename(trainer.checkpoint_path) == 'STC-GNN-4.pkl'
    res = trainer.test(loaders)
    assert res['test']['forecast'].shape == (loaders['test'].length, 2, 6, 2) and 0 < res['test']['bce'] < 2
    if 'metrics' in res['test']:                                # the reference's evaluation protocol, one dict per horizon step
        assert len(res['test']['metrics']) == 2 and set(res['test']['metrics'][0]) >= {'Macro-F1', 'Micro-F2', 'ROC-AUC', 'PR-AUC', 'BCE', 'MAE'}


def test_trainer_refuses_cpu_device():
    data, params = pipeline_inputs()
    with pytest.raises(ValueError, match='no CPU implementation'):
        Trainer(dict(params, output_dir='/tmp'), data)


@pytest.mark.gpu
def test_trainer_on_the_gpu_reproduces_the_reference_epoch_losses(tmp_path, monkeypatch):
    """The same 2-epoch run through libstc_hip.so on the MI355X (learned dense graphs, MGP_Gen included)."""
    monkeypatch.setattr(ops, '_kernels', None)
    g = load_golden('g9_pipeline')
    data, params = pipeline_inputs()
    params = dict(params, device='cuda:0', output_dir=str(tmp_path))
    loaders = sdata.get_data_loader(params, data, params['obs_len'], params['pred_len'], params['split_ratio'])
    torch.manual_seed(123)
    trainer = Trainer(params, data)
    hist = trainer.train(loaders, verbose=False)
    assert np.allclose(hist['loss']['train'], g['train_curve'].numpy(), rtol=0, atol=5e-5), hist['loss']
    assert np.allclose(hist['loss']['validate'], g['val_curve'].numpy(), rtol=0, atol=5e-5), hist['loss']
    assert trainer.test(loaders)['test']['forecast'].shape[0] == loaders['test'].length


@pytest.mark.gpu
@pytest.mark.parametrize('C', [5, 32])       # 5: one autograd node per cell (generic kernels); 32: the planar cell graph
def test_hip_graph_replay_of_a_train_step_equals_eager(monkeypatch, C):
    """The whole step (forward, ComboLoss, backward, Adam) captured with torch.cuda.graph: every launch goes to the
    capturing stream and allocates only through torch, so replay must reproduce the eager parameters bit for bit."""
    import STC_GNN as M
    from stc_hip import CsrGraph
    from stc_hip.loss import ComboLoss
    monkeypatch.setattr(ops, '_kernels', None)
    dev = torch.device('cuda')

    def build():
        torch.manual_seed(5)
        model = M.STCGNN(64, C, 2, 2, 1, 16, 2, 2, graph_mode='csr-fixed').to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=2e-3, weight_decay=1e-4, capturable=True)
        return model, opt

    graph = CsrGraph.queen_grid(8, 8, device=dev)
    gen = torch.Generator().manual_seed(1)
    X = (torch.rand(4, 3, 64, C, generator=gen) < 0.3).float().to(dev)
    Y = (torch.rand(4, 2, 64, C, generator=gen) < 0.3).float().to(dev)
    Gc = torch.softmax(torch.randn(C, C, generator=gen), -1).to(dev)
    crit = ComboLoss()

    def make_step(model, opt):
        def step():
            opt.zero_grad(set_to_none=True)
            loss = crit(model(X_seq=X, As=graph, Ac=Gc), Y)
            loss.backward()
            opt.step()
            return loss
        return step

    m1, o1 = build()
    s1 = make_step(m1, o1)
    for _ in range(5):
        s1()
    m2, o2 = build()
    s2 = make_step(m2, o2)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            s2()                                   # warm-up steps count: 2 eager + 1 capture-less... see below
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s2()                                       # capture does not execute
    for _ in range(3):
        g.replay()                                 # 2 eager + 3 replays = 5 steps
    torch.cuda.synchronize()
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize('where', ['numpy', 'cpu', pytest.param('cuda', marks=pytest.mark.gpu)])
def test_metrics_match_the_reference(where):
    """mask_data and the per-step metric dict equal the reference's (scikit-learn based) values; the work
    runs on the device the predictions live on."""
    from stc_hip import metrics as smetrics
    from tests.golden.make_golden import metrics_inputs
    g = load_golden('g10_metrics')
    prob, true, thr, mask, H, W = metrics_inputs()
    conv = (lambda a: a) if where == 'numpy' else (lambda a: torch.from_numpy(a).to(where))
    host = (lambda a: a) if where == 'numpy' else (lambda a: a.cpu().numpy())
    pm, tm = smetrics.mask_data(conv(prob), H, W, mask), smetrics.mask_data(conv(true), H, W, mask)
    if where != 'numpy':
        assert pm.device.type == where
    assert np.array_equal(host(pm), g['prob_masked'].numpy()) and np.array_equal(host(tm), g['true_masked'].numpy())
    assert np.array_equal(host(smetrics.mask_data(conv(prob), H, W, None)), g['unmasked'].numpy())
    grid = prob.reshape(prob.shape[0], prob.shape[1], H, W, -1).transpose(0, 1, 4, 2, 3)      # (S, hor, C, H, W) layout
    assert np.array_equal(host(smetrics.mask_data(conv(np.ascontiguousarray(grid)), H, W, mask)), host(pm))
    steps = smetrics.evaluate_binary(pm, tm, list(thr))
    for s, got in enumerate(steps):
        assert list(got.keys()) == list(g[f'names{s}'])
        assert np.allclose(list(got.values()), g[f'values{s}'].numpy(), rtol=0, atol=1.01e-4), (got, g[f'values{s}'])


def _run_bench(extra, env=None, timeout=900):
    import json
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '2', '--warmup', '1', '--batch-per-gpu', '2', '--obs', '3', '--pred', '2'] + extra
    if '--presets' in cmd:
        cmd.remove('--presets')
    else:
        cmd.append('--no-presets')                          # (the presets block has a test of its own: six child processes)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=None if env is None else {**os.environ, **env})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().split('\n') if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize('storage', ['f32', 'bf16'])
def test_bench_line_contract(storage):
    """bench.py prints ONE JSON line with the driver's contract fields, the roofline object of the plain aggregation launches
    (SURVEY 8(d3) unit beside it) and (fp32, one GPU) the CPU baseline; run here on a small grid so that it takes seconds."""
    d = _run_bench(['--gpus', '1', '--grid', '12', '--storage', storage] + (['--no-cpu-baseline'] if storage == 'bf16' else []),
                   env={'STC_BENCH_NO_DENSE_ANCHOR': '1'})
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
                'data', 'config', 'roofline', 'n_ranks_seen', 'step_breakdown'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['unit'] == 'samples/s' and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['data'] == 'synthetic' and d['dtype'] == storage
    assert d['value'] > 0 and abs(d['value'] - 2 * 1e3 / d['ms_per_step']) < 1e-6 * d['value'] and d['n_ranks_seen'] == 1
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and r['launches'] > 0 and 'spmm_bcsr_kernel' in r['kernel']
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    N, nnz, C, B = 144, 4 * 3 + 40 * 5 + 100 * 8, 32, 2                      # 12 x 12 queen grid
    esize = 2 if storage == 'bf16' else 4
    if storage == 'f32':                 # (bf16 also sends its narrow layer-0 input planes, F = C, through the row-blocked kernel)
        assert r['algorithmic_bytes_per_launch'] == nnz * 8 + 4 * (N + 1) + 2 * B * N * C * 16 * esize     # SURVEY 8(d3), verbatim
    assert r['traffic'] is None and r['traffic_source'] is None              # no PMC pass over this configuration: never a stale number
    dom = r['dominant']
    assert dom['entry_point'] in d['kernels'] and 0 < dom['share_of_kernel_time'] <= 1 and dom['launches'] > 0
    assert r['mfma'] is None or r['mfma'].get('value', 0) is None or r['mfma']['source'] == 'committed'     # quoted only for the same kernel sources
    u = r['unit_d3']
    assert u['algorithmic_bytes'] == nnz * 8 + 4 * (N + 1) + 2 * N * C * 32 * esize and u['avg_launch_us'] > 0
    assert r['aggregate']['launches'] >= r['launches'] and r['aggregate']['achieved'] > 0
    pw = d['power']                       # package power / shader clock over the timed region (amdsmi), or null where it cannot be read
    assert pw is None or (pw['samples'] > 0 and pw['package_w_mean'] > 0 and 0 < pw['sclk_mhz_min'] <= pw['sclk_mhz_mean'])
    sb = d['step_breakdown']
    assert sb['fwd_loss_bwd_ms'] > 0 and sb['adam_ms'] > 0 and sb['ms_per_step_without_launch_events'] > 0
    assert sb['fwd_loss_bwd_ms'] + sb['grad_allreduce_ms'] + sb['adam_ms'] <= d['ms_per_step'] * 1.05
    names = set(d['kernels'])
    assert any(n.endswith('_bf16') for n in names) == (storage == 'bf16')
    if storage == 'f32':
        c = d['cpu_baseline']
        assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == 'samples/s' and 'sample' in c
        # a small graph (here 12 x 12): the oracle's WHOLE model, dense reference algorithm, 2 warm-up + 7 timed steps; the
        # metric's size takes one cell of each kind instead (shots_s = {'layer0': [...], 'wide': [...]})
        assert len(c['shots_s']) == 9 and 'whole model' in c['sample']
    assert d['per_rank']['ms_per_step_min'] <= d['per_rank']['ms_per_step_max'] and len(d['per_rank']['ms_per_step']) == 1 and d['per_rank']['build_s_max'] > 0


@pytest.mark.gpu
def test_bench_presets_block_and_fp32_grade_value():
    """The default one-GPU line carries (a) ``value_fp32_grade``: the same step on the 24-bit operand format over the SAME --steps as ``value``, and
    (b) ``presets``: BASELINE.json's other configurations, each a child process run after the metric's measurements (here two of them, by name)."""
    d = _run_bench(['--gpus', '1', '--grid', '12', '--no-cpu-baseline', '--no-unit-d3', '--presets'], env={'STC_BENCH_PRESETS': 'cfg2,sf'})
    assert d['dtype'] == 'f32' and d['config']['operand_format'] == 'f16x2' and 'two fp16 pieces' in d['dtype_detail']
    g = d['value_fp32_grade']
    assert g['operand_format'] == 'bf16x3' and g['steps'] == d['steps'] == 2 and g['value'] > 0 and g['unit'] == 'samples/s'
    assert abs(g['value'] - d['alt_formats']['bf16x3']['samples_per_s']) < 1e-9
    p = d['presets']
    assert p['parent_reserved_gb_while_they_ran'] < 1.0 and set(p) - {'what', 'parent_reserved_gb_while_they_ran'} == {'cfg2', 'sf'}
    for name, unit in (('cfg2', 'layers/s'), ('sf', 'samples/s')):
        assert 'error' not in p[name], p[name]
        assert p[name]['unit'] == unit and p[name]['value'] > 0 and p[name]['steps'] >= 3 and p[name]['roofline']['frac'] > 0
    assert p['sf']['hip_graph'] is True and p['sf']['roofline']['dominant']['entry_point'].startswith('stc_cell_small')


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """``python bench.py --gpus 2`` with no launcher environment (the driver's plain form) starts torch.distributed.run as a
    child, both ranks take part in the collectives, rank 0 prints the one line.  A one-GPU box cannot run RCCL between two
    ranks, so the ranks share device 0 and talk over gloo (test hooks of stc_hip.dist.init_from_env); the launch path,
    sharding, barrier / max-over-ranks timing and tear-down are the ones an 8-GPU RCCL run takes."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    import json
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--grid', '32', '--batch-per-gpu', '2',
           '--obs', '3', '--pred', '2']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env={**env, 'STC_DIST_BACKEND': 'gloo', 'STC_DIST_ONE_DEVICE': '1'})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().split('\n') if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['n_ranks_seen'] == 2 and d['config']['global_batch'] == 4 and d['config']['parallelism'] == 'batch-shard x2'
    assert abs(d['value'] - 4 * 1e3 / d['ms_per_step']) < 1e-6 * d['value'] and 'cpu_baseline' not in d
    assert d['step_breakdown']['grad_allreduce_ms'] > 0


@pytest.mark.gpu
def test_bench_many_ranks_contract():
    """The driver's multi-GPU line, rehearsed at the largest rank count a one-GPU box allows (four ranks + their launcher + this process = the six
    processes its guard admits -- five ranks were killed by it; the ranks share device 0 and talk over gloo): every rank is seen by a collective, ``per_rank`` holds one step time per rank,
    ``ms_per_step`` -- hence ``value`` -- is the SLOWEST rank's, the global batch is ranks x batch-per-gpu, and one JSON line comes out.  Nothing in
    bench.py depends on the rank count beyond these: the first real 8-GPU run takes the same path with ``nccl`` (RCCL) as the backend."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    import json
    import subprocess
    import sys
    n = 4
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(n), '--steps', '3', '--warmup', '1', '--grid', '32', '--batch-per-gpu', '2',
           '--obs', '3', '--pred', '2']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env={**env, 'STC_DIST_BACKEND': 'gloo', 'STC_DIST_ONE_DEVICE': '1'})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().split('\n') if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == n and d['n_ranks_seen'] == n and d['config']['global_batch'] == 2 * n and d['config']['parallelism'] == f'batch-shard x{n}'
    pr = d['per_rank']
    assert len(pr['ms_per_step']) == n and abs(pr['ms_per_step_max'] - max(pr['ms_per_step'])) < 1e-3 and abs(pr['ms_per_step_min'] - min(pr['ms_per_step'])) < 1e-3
    assert abs(d['ms_per_step'] - pr['ms_per_step_max']) < 1e-2 * d['ms_per_step'] + 1e-3          # the slowest rank's clock (list entries are rounded)
    assert abs(d['value'] - 2 * n * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    assert d['scaling'] == 'weak' and 'cpu_baseline' not in d and 'alt_formats' not in d and 'permuted' not in d
    assert 'presets' not in d and 'value_fp32_grade' not in d             # side measurements and child processes: one GPU, rank 0 only


def _two_gpus():
    """RCCL between ranks needs one GPU per rank: these tests run on a node with >= 2 visible GPUs and say so loudly otherwise."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f'RCCL between two ranks needs two GPUs; {n} visible on this box (the gloo / one-device form of the same path runs in '
                    'test_bench_launches_its_own_ranks and test_batch_sharded_trainer_on_the_gpu_follows_the_reference_curves)')


@pytest.mark.gpu
def test_bench_two_ranks_over_rccl():
    """``python bench.py --gpus 2`` with the DEFAULT backend (nccl = RCCL over xGMI), one rank per GPU: both ranks take part in the
    collectives, and rank 0's loss equals the gloo run's (same shards, same parameters: the backend must not change the arithmetic).
    Strong scaling form (``--global-batch``) as well."""
    _two_gpus()
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'STC_DIST_BACKEND', 'STC_DIST_ONE_DEVICE')}
    base = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--grid', '32', '--obs', '3', '--pred', '2']
    got = {}
    for name, extra, e in (('rccl', ['--batch-per-gpu', '2'], env), ('gloo', ['--batch-per-gpu', '2'], {**env, 'STC_DIST_BACKEND': 'gloo'}),
                           ('strong', ['--global-batch', '4'], env)):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=900, env=e)
        assert out.returncode == 0, (name, out.stderr[-3000:])
        got[name] = json.loads([l for l in out.stdout.strip().split('\n') if l.startswith('{')][0])
    for name, d in got.items():
        assert d['n_gpus'] == 2 and d['n_ranks_seen'] == 2 and d['config']['global_batch'] == 4, name
        assert d['step_breakdown']['grad_allreduce_ms'] > 0
    assert abs(got['rccl']['loss'] - got['gloo']['loss']) < 1e-6 and abs(got['strong']['loss'] - got['rccl']['loss']) < 1e-6
    assert got['rccl']['scaling'] == 'weak' and got['strong']['scaling'] == 'strong'


@pytest.mark.gpu
def test_batch_sharded_trainer_over_rccl_follows_the_reference_curves(tmp_path):
    """The trainer counterpart under ``torch.distributed.run``, two ranks on two GPUs over RCCL (default backend): parameters broadcast
    from rank 0, learned graphs with the batch-sum all-reduce, ragged last batch, one gradient-bucket all-reduce per step -- the
    REFERENCE's 2-epoch loss curves of g9."""
    _two_gpus()
    import json
    import subprocess
    import sys
    script = tmp_path / 'sharded_trainer.py'
    script.write_text(_SHARDED_TRAINER_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'STC_DIST_BACKEND', 'STC_DIST_ONE_DEVICE')}
    env.update(STC_REPO=REPO, STC_OUT=str(tmp_path))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', '29549', str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.strip().split('\n') if l.startswith('{')][0])
    g = load_golden('g9_pipeline')
    assert d['world'] == 2
    assert np.allclose(d['train'], g['train_curve'].numpy(), rtol=0, atol=1e-4), d
    assert np.allclose(d['val'], g['val_curve'].numpy(), rtol=0, atol=1e-4), d


@pytest.mark.gpu
@pytest.mark.parametrize('K', [2, 3])
def test_bench_preset_cfg2(K):
    """bench.py --preset cfg2: the single BDG_Dif layer of BASELINE configuration 2, forward and forward + backward, beside the reference's CPU times."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--preset', 'cfg2', '--order', str(K)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.strip().split('\n') if l.startswith('{')][0])
    assert d['config']['preset'] == 'cfg2' and d['forward_ms'] > 0 and d['forward_backward_ms'] > 0      # (timings of a 0.1-1 ms launch: no ordering asserted)
    assert d['reference_cpu_ms']['forward'] == {2: 13.3, 3: 29.6}[K] and d['vs_baseline'] is None


@pytest.mark.gpu
@pytest.mark.parametrize('preset', ['sf', 'sf-learned'])
def test_bench_preset_sf(preset):
    """bench.py --preset sf / sf-learned: the SF-incidents shape (fixed sparse graph / the reference's full model with MGP_Gen's learned dense
    graphs) runs on the few-category cell kernels, replayed as a HIP graph, and the line prices them against the fp32 matrix peak."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--preset', preset, '--steps', '3', '--warmup', '2', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.strip().split('\n') if l.startswith('{')][0])
    r = d['roofline']
    assert d['hip_graph'] is True and d['config']['preset'] == preset and d['config']['global_batch'] == 32
    assert ('dense-learned' in d['config']['workload']) == (preset == 'sf-learned')
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and 0 < r['frac'] < 1 and r['dominant']['entry_point'].startswith('stc_cell_small')
    m = r['dominant']['matrix']
    assert m['launches_per_cell_step'] in (1, 2, 3, 4) and m['compute_units_in_use'] in (32, 256)
    assert set(d['kernels']) >= {'stc_cell_small_fwd_f32', 'stc_cell_small_bwd_f32'}
    if preset == 'sf-learned':
        assert {'stc_graph_grad_f32', 'stc_mix_dt_f32'} <= set(d['kernels'])      # (dGs pieces; dT_c of every parameter set on the matrix cores)


@pytest.mark.gpu
def test_bench_under_torchrun_single_rank_goes_through_rccl():
    """The driver's N > 1 form with N = 1: torch.distributed.run sets RANK, so the process group IS initialised (backend
    nccl = RCCL) and the barrier, the max-over-ranks all-reduce and destroy_process_group run on the real communicator."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', '29533',
           os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--grid', '12', '--batch-per-gpu', '2', '--obs', '3', '--pred', '2',
           '--no-cpu-baseline']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.strip().split('\n') if l.startswith('{')][0])
    assert d['n_gpus'] == 1 and d['n_ranks_seen'] == 1


_SHARDED_TRAINER_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, os.environ['STC_REPO']); sys.path.insert(0, os.path.join(os.environ['STC_REPO'], 'stc-gnn_amd'))
import torch
from stc_hip import data as sdata, dist as sdist
from stc_hip.trainer import Trainer
from tests.golden.make_golden import pipeline_inputs
rank, world, local = sdist.init_from_env()
data, params = pipeline_inputs()
params = dict(params, device=f'cuda:{local}', output_dir=os.environ['STC_OUT'])
loaders = sdata.get_data_loader(params, data, params['obs_len'], params['pred_len'], params['split_ratio'])
torch.manual_seed(123 if rank == 0 else 1000 + rank)      # the other ranks must receive rank 0's parameters (broadcast in Trainer)
trainer = Trainer(params, data)
hist = trainer.train(loaders, verbose=False)
flat = torch.cat([p.detach().flatten() for p in trainer.model.parameters()]).double()
if rank == 0:
    print(json.dumps(dict(world=trainer.world, train=hist['loss']['train'], val=hist['loss']['validate'], checksum=float(flat.sum()))))
torch.distributed.barrier(); torch.distributed.destroy_process_group()
'''


@pytest.mark.gpu
def test_batch_sharded_trainer_on_the_gpu_follows_the_reference_curves(tmp_path):
    """The trainer counterpart under ``torch.distributed.run`` with two ranks through libstc_hip.so (learned graphs with the batch-sum
    all-reduce, ragged last batch, one gradient-bucket all-reduce per step): the REFERENCE's 2-epoch loss curves of g9.  The box has one
    GPU, so both ranks sit on it and talk over gloo (the hooks of stc_hip.dist.init_from_env); on a node this is RCCL, one rank per GPU."""
    import json
    import subprocess
    import sys
    script = tmp_path / 'sharded_trainer.py'
    script.write_text(_SHARDED_TRAINER_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(STC_REPO=REPO, STC_OUT=str(tmp_path), STC_DIST_BACKEND='gloo', STC_DIST_ONE_DEVICE='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', '29547', str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.strip().split('\n') if l.startswith('{')][0])
    g = load_golden('g9_pipeline')
    assert d['world'] == 2
    assert np.allclose(d['train'], g['train_curve'].numpy(), rtol=0, atol=1e-4), d
    assert np.allclose(d['val'], g['val_curve'].numpy(), rtol=0, atol=1e-4), d
    assert os.path.exists(tmp_path / 'STC-GNN-4.pkl')


def test_synthetic_dataset_follows_the_reference_schema():
    d = sdata.synthetic_incidents(10, 10, 5, 40)
    assert set(d) == {'inc', 'mask', 'HA', 's_adj', 'c_cor'} and d['inc'].shape == (40, 10, 10, 5) and d['inc'].dtype == np.int32
    assert d['s_adj'].shape == (100, 100) and d['s_adj'].sum() == 684 and (d['s_adj'] == d['s_adj'].T).all()      # the SF file's s_adj
    assert d['c_cor'].shape == (5, 5) and d['HA'].shape == (5,) and 0.1 < d['HA'].mean() < 0.23
    from stc_hip import CsrGraph
    assert isinstance(sdata.synthetic_incidents(4, 5, 3, 12, sparse_graph=True)['s_adj'], CsrGraph)


@pytest.mark.gpu
def test_train_cli_on_a_synthetic_grid_with_a_fixed_sparse_graph(tmp_path):
    """tools/train.py (Main.py's flags + -synthetic / -graph): two epochs of csr-fixed training at N = 1 600, C = 32 through the planar
    cell graph, best-validation checkpoint, test from the checkpoint, the metrics CSV."""
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(REPO, 'tools', 'train.py'), '-synthetic', '40', '40', '32', '44', '-graph', 'csr-fixed', '-epoch', '2',
           '-batch', '4', '-obs', '4', '-pred', '2', '-out', str(tmp_path)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    losses = [float(v) for v in re.findall(r'Epoch \d+: train ([0-9.]+)', out.stdout)]
    assert len(losses) == 2 and losses[1] < losses[0], out.stdout
    assert os.path.exists(tmp_path / 'synthetic' / 'STC-GNN-4.pkl') and os.path.exists(tmp_path / 'synthetic' / 'STC-GNN_eval-bi-metrics.csv')
    assert "'test'" in out.stdout and 'bce' in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['dense-learned', 'csr-fixed'])
def test_trainer_with_hip_graph_replay_equals_eager(tmp_path, monkeypatch, mode):
    """Trainer(hip_graph=True): after two eager steps per batch shape the whole step is replayed from a captured HIP graph.  Same
    epoch losses as the eager trainer (the learned-graph run: the reference's g9 curves) and the same final parameters."""
    monkeypatch.setattr(ops, '_kernels', None)
    g = load_golden('g9_pipeline')
    data, params = pipeline_inputs()
    params = dict(params, device='cuda:0', num_epochs=3)
    if mode == 'csr-fixed':
        from stc_hip import CsrGraph
        data = dict(data, s_adj=CsrGraph.queen_grid(params['H'], params['W'], normalize=True))
    hist, flat = {}, {}
    for graphed in (False, True):
        out = tmp_path / str(graphed)
        os.makedirs(out)
        loaders = sdata.get_data_loader(params, data, params['obs_len'], params['pred_len'], params['split_ratio'])
        torch.manual_seed(123)
        trainer = Trainer(dict(params, output_dir=str(out)), data, graph_mode=mode, hip_graph=graphed)
        hist[graphed] = trainer.train(loaders, verbose=False)['loss']
        flat[graphed] = torch.cat([p.detach().flatten() for p in trainer.model.parameters()])
        if graphed:
            assert any(slot[1] is not None for slot in trainer._captured.values()), 'no step was captured'
    assert np.allclose(hist[True]['train'], hist[False]['train'], rtol=0, atol=1e-6), hist
    assert np.allclose(hist[True]['validate'], hist[False]['validate'], rtol=0, atol=1e-6), hist
    assert float((flat[True] - flat[False]).abs().max()) < 1e-6
    if mode == 'dense-learned':
        assert np.allclose(hist[True]['train'][:2], g['train_curve'].numpy(), rtol=0, atol=5e-5)


@pytest.mark.gpu
def test_bench_hip_graph_option():
    """bench.py --hip-graph: `value` over replays of the captured train step; the per-kernel events come from eager steps afterwards."""
    d = _run_bench(['--gpus', '1', '--grid', '12', '--hip-graph', '--no-cpu-baseline'])
    assert d['hip_graph'] is True and d['value'] > 0 and d['roofline']['launches'] > 0 and d['roofline']['achieved'] > 0
    assert abs(d['value'] - 2 * 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    assert 0.5 < d['loss'] < 2.0 and set(d['kernels']) >= {'stc_bcsr_spmm_f32', 'stc_cell_gates_fwd_planar_f32'}      # (more steps have run than in the eager line)
