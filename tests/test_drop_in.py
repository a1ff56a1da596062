"""SURVEY section 4, "drop-in" tier: the reference's OWN caller stack on top of the build's ``STC_GNN`` module.

``framework/Model_Trainer.py:5`` does ``from STC_GNN import STCGNN``; with ``stc-gnn_amd/`` ahead of ``framework/`` on
``sys.path`` that import resolves to the build's module, and the reference's ``ModelTrainer`` (``:26-158``),
``Data_Container.DataGenerator`` and ``Metrics.ModelEvaluator`` run UNCHANGED on it: constructor keywords (``:39-46``),
``forward(X_seq=, As=, Ac=)`` (``:74, :138``), ``.parameters()`` into Adam (``:35``), ``.train()/.eval()``,
``state_dict()`` into the checkpoint and ``load_state_dict()`` back (``:53, :103, :126-128``).

Build container only (the reference does not travel to the GPU box): skipped when ``/root/reference`` is absent.  Compute
runs on the emulated kernel set (no GPU here), injected as in the other CPU tests; the GPU run of the same loop through
libstc_hip.so is ``test_pipeline.py::test_trainer_on_the_gpu_reproduces_the_reference_epoch_losses``.
"""
import contextlib
import importlib.util
import io
import os
import re
import sys

import numpy as np
import pytest
import torch

from oracle.kernel_emul import EmulatedKernels
from stc_hip import ops
from tests.conftest import PKG, REFERENCE, load_golden, sub_dict
from tests.golden.make_golden import pipeline_inputs

pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason='reference not present (build container only)')


def _import_reference_callers():
    """The reference's Model_Trainer / Data_Container / Metrics, imported with the BUILD's STC_GNN first on sys.path."""
    sys.dont_write_bytecode = True
    saved_path = list(sys.path)
    saved_mods = {k: sys.modules.get(k) for k in ('Model_Trainer', 'Data_Container', 'Metrics')}
    sys.path[:] = [PKG] + [p for p in sys.path if p not in (PKG, REFERENCE)] + [REFERENCE]      # build first, reference last
    try:
        mods = {}
        for name in ('Data_Container', 'Model_Trainer'):
            sys.modules.pop(name, None)
            spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE, name + '.py'))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)                          # its "from STC_GNN import STCGNN" runs here
            mods[name] = mod
        return mods['Model_Trainer'], mods['Data_Container']
    finally:
        sys.path[:] = saved_path
        for k, v in saved_mods.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_reference_model_trainer_runs_on_the_drop_in_module(tmp_path, monkeypatch):
    import STC_GNN as M
    MT, DC = _import_reference_callers()
    assert MT.STCGNN is M.STCGNN and os.path.realpath(sys.modules[MT.STCGNN.__module__].__file__).startswith(os.path.realpath(PKG)), \
        'Model_Trainer.py:5 did not resolve to the build\'s STC_GNN'
    monkeypatch.setattr(ops, '_kernels', EmulatedKernels())
    g = load_golden('g9_pipeline')
    data, params = pipeline_inputs()
    params = dict(params, output_dir=str(tmp_path))
    gen = DC.DataGenerator(obs_len=params['obs_len'], pred_len=params['pred_len'], data_split_ratio=params['split_ratio'])
    loaders = gen.get_data_loader(params=params, data=data)                      # the reference's own DataLoader objects
    torch.manual_seed(123)
    trainer = MT.ModelTrainer(params=params, data=data)                          # Model_Trainer.py:26-46, unchanged
    assert isinstance(trainer.model, M.STCGNN)
    for k, v in sub_dict(g, 'sd0/').items():                                     # same seed -> the reference's initial parameters
        assert torch.equal(trainer.model.state_dict()[k], v), k
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        trainer.train(data_loader=loaders, modes=['train', 'validate'])           # Model_Trainer.py:52-121, unchanged
    printed = buf.getvalue()
    train_printed = [float(v) for v in re.findall(r'training loss: ([0-9.]+)', printed)]
    val_printed = [float(v) for v in re.findall(r'to ([0-9.]+)\. Update model checkpoint', printed)]
    want_train, want_val = g['train_curve'].numpy(), g['val_curve'].numpy()
    assert len(train_printed) == params['num_epochs']
    assert np.allclose(train_printed, want_train, rtol=6e-4, atol=0), (train_printed, want_train)      # 4 printed digits
    ck = torch.load(os.path.join(str(tmp_path), 'STC-GNN-4.pkl'), weights_only=False)
    assert sorted(ck.keys()) == list(g['ckpt_keys']) and ck['epoch'] == int(g['ckpt_epoch'])
    e = ck['epoch'] - 1                                                           # the checkpoint holds that epoch's exact losses
    assert abs(float(ck['train_loss']) - want_train[e]) < 2e-5 and abs(float(ck['val_loss']) - want_val[e]) < 2e-5
    assert val_printed and abs(val_printed[-1] - want_val[e]) < 6e-4 * want_val[e]
    assert list(ck['state_dict'].keys()) == list(trainer.model.state_dict().keys())
    with contextlib.redirect_stdout(io.StringIO()):
        trainer.test(data_loader=loaders, modes=['test'])                         # :124-158: load_state_dict, forward, Metrics
    csv = os.path.join(str(tmp_path), 'STC-GNN_eval-bi-metrics.csv')
    assert os.path.exists(csv) and 'Macro-F1' in open(csv).read()
    # the build's trainer counterpart writes the same log: same file name, same line structure, same header row, same values
    from stc_hip import data as sdata
    from stc_hip.trainer import Trainer
    ref_lines = open(csv).read().split('\n')
    os.remove(csv)
    mine = Trainer(dict(params, _allow_cpu_for_tests=True), data)
    mine.test(sdata.get_data_loader(params, data, params['obs_len'], params['pred_len'], params['split_ratio']))
    my_lines = open(csv).read().split('\n')
    assert len(my_lines) == len(ref_lines)
    scrub = lambda l: re.sub(r'(starts|ends), (\w+), [^,]*,', r'\1, \2, TIME,', l)
    for a, b in zip(my_lines, ref_lines):
        if a.startswith('*****'):
            assert scrub(a) == scrub(b)
        elif a.startswith('device:') or 'device:' in a[:40]:
            assert a == b                                            # the parameter dump
        elif a.startswith('Step'):
            assert a.split(',')[0] == b.split(',')[0]
            assert np.allclose([float(v) for v in a.split(',')[1:]], [float(v) for v in b.split(',')[1:]], atol=1.01e-4)
        else:
            assert a == b                                            # header row, blank lines


MAIN_SLICES = 320      # time slices of the SF file the default run keeps (STC_MAIN_FULL_FILE=1: all 5 124, ~12 min on 8 cores through the emulated kernels)


def test_reference_main_runs_on_the_drop_in_module(tmp_path, monkeypatch):
    """BASELINE.json configuration 1, literally: the reference's ``Main.py`` (``:64-80``) as a script (``runpy.run_path``) on the SF file, with
    the build's ``STC_GNN`` first on ``sys.path``: ``-device cpu -city SF -split 1 3 3 -epoch 1`` (the reduced split is SURVEY F8: the reference's
    trainer keeps every step's autograd graph alive, the default split does not fit this container).  No GPU here, so the kernels are the
    emulated set (80 ms per sample forward): by default ``-in`` points at a copy of the SF file cut to its first MAIN_SLICES time slices -- same
    schema, same graphs, ~1 min; with STC_MAIN_FULL_FILE=1 at the reference's own file.  What is checked is the plumbing ``Main.py`` exercises --
    argument parsing, ``DataInput`` on the ``.npz``, the loaders, ``ModelTrainer`` building the drop-in ``STCGNN`` from ``params``, one epoch of
    train + validate, the checkpoint, ``test`` and its metrics file."""
    import runpy
    import STC_GNN as M
    data_dir = os.path.join(os.path.dirname(REFERENCE), 'data')
    if not os.path.exists(os.path.join(data_dir, 'SF-incidents-4h.npz')):
        pytest.skip('the SF data file is not present')
    full = os.environ.get('STC_MAIN_FULL_FILE') == '1'
    if not full:
        with np.load(os.path.join(data_dir, 'SF-incidents-4h.npz'), allow_pickle=True) as z:
            cut = {k: (z[k][:MAIN_SLICES] if k in ('incident', 'metadata') else z[k]) for k in z.files}
        data_dir = str(tmp_path / 'data')
        os.makedirs(data_dir)
        np.savez(os.path.join(data_dir, 'SF-incidents-4h.npz'), **cut)
    monkeypatch.setattr(ops, '_kernels', EmulatedKernels())
    monkeypatch.setattr(sys, 'path', [PKG] + [p for p in sys.path if p not in (PKG, REFERENCE)] + [REFERENCE])     # build first, reference last
    for name in ('Model_Trainer', 'Data_Container', 'Metrics'):
        monkeypatch.delitem(sys.modules, name, raising=False)
    monkeypatch.setattr(sys, 'dont_write_bytecode', True)
    monkeypatch.setattr(sys, 'argv', ['Main.py', '-device', 'cpu', '-in', data_dir, '-out', str(tmp_path), '-city', 'SF',
                                      '-split', '1', '3', '3', '-epoch', '1'])
    torch.manual_seed(0)
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            ns = runpy.run_path(os.path.join(REFERENCE, 'Main.py'), run_name='__main__')
    finally:
        for name in ('Model_Trainer', 'Data_Container', 'Metrics'):
            sys.modules.pop(name, None)
    printed = buf.getvalue()
    trainer = ns['trainer']
    assert type(trainer.model) is M.STCGNN, 'Model_Trainer.py:5 did not resolve to the build\'s STC_GNN'
    assert ns['params']['C'] == 5 and (ns['params']['H'], ns['params']['W']) == (10, 10)
    assert ns['data']['inc'].shape == ((5124 if full else MAIN_SLICES), 10, 10, 5)
    n_windows = ns['data']['inc'].shape[0] - 12
    assert len(ns['data_loader']['train'].dataset) == n_windows - 2 * int(3 / 7 * n_windows)
    losses = [float(v) for v in re.findall(r'Epoch \d+: .*?training loss: ([0-9.]+)', printed)]
    assert len(losses) == 1 and 0.5 < losses[0] < 1.6 and 'Successfully loaded trained STC-GNN model - epoch: 1' in printed, printed[:2000]          # (the reference itself, seed 0, SF defaults: 1.46 falling to 1.28 in 12 steps)
    ck = torch.load(os.path.join(str(tmp_path), 'SF', 'STC-GNN-4.pkl'), weights_only=False)
    assert sorted(ck.keys()) == ['epoch', 'state_dict', 'train_loss', 'val_loss'] and ck['epoch'] == 1
    assert list(ck['state_dict'].keys()) == list(trainer.model.state_dict().keys()) and len(ck['state_dict']) == 32
    csv = os.path.join(str(tmp_path), 'SF', 'STC-GNN_eval-bi-metrics.csv')
    assert os.path.exists(csv) and 'Macro-F1' in open(csv).read()
