"""Host logic of bench.py that needs no GPU: the self-launch of ``--gpus N`` and the guards around ``roofline.traffic``."""
import json
import os
import sys

import pytest

from tests.conftest import REPO

sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_gpus_n_without_a_launcher_starts_torchrun_as_a_child(monkeypatch):
    """``python bench.py --gpus 4`` (no RANK in the environment): torch.distributed.run is started as a child process with the
    driver's argument form, the original flags are passed through, and the exit code is the child's."""
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '20', '--warmup', '5'])
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run'] and '--nnodes=1' in cmd and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    i = cmd.index(os.path.join(REPO, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '4', '--steps', '20', '--warmup', '5']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_traffic_is_quoted_only_for_matching_sources_and_configuration(tmp_path, monkeypatch):
    key = 'f32:224:32:16:5:2:2:18:6:0'
    doc = dict(csrc_sha=bench.csrc_sha(), config_key=key,
               kernels={'spmm_bcsr_kernel<1, 0, 2, 1, 1>': dict(hbm_bytes_per_launch=100.0, launches=3),
                        'spmm_bcsr_kernel<1, 3, 2, 0, 0>': dict(hbm_bytes_per_launch=900.0, launches=5),       # a sum form: not the plain launch
                        'node_fwd_x3_kernel<1, 2, 2, 32, 1, 0, 1, 1>': dict(hbm_bytes_per_launch=5.0, launches=1)})
    monkeypatch.setattr(bench, 'REPO', str(tmp_path))
    os.makedirs(tmp_path / 'profiles' / 'r03')
    monkeypatch.setattr(bench, 'csrc_sha', lambda: doc['csrc_sha'])
    path = tmp_path / 'profiles' / 'r03' / 'hbm_traffic_bench.json'
    path.write_text(json.dumps(doc))
    t, note = bench.pmc_traffic(None, key)
    assert t == 100.0 and 'profiles/r03/hbm_traffic_bench.json' in note
    assert bench.pmc_traffic(None, 'f32:100:32:16:4:3:2:18:6:0')[0] is None            # another configuration
    path.write_text(json.dumps(dict(doc, csrc_sha='0' * 16)))
    t, note = bench.pmc_traffic(None, key)
    assert t is None and 'not quoted' in note                                           # other kernel sources: never a stale number
    os.remove(path)
    assert bench.pmc_traffic(None, key) == (None, None)


def test_matrix_pipe_counters_are_quoted_only_for_matching_sources(tmp_path, monkeypatch):
    key = 'f32:224:32:16:5:2:2:18:6:0'
    doc = dict(csrc_sha=bench.csrc_sha(), config_key=key, definition='...',
               kernels={'cell_bwd_x3_kernel<FmtH2, 32, 1, 0, 0>': dict(launches=48, mfma_util=0.3, valu_busy=0.4),
                        'spmm_bcsr_kernel<1, 0, 2, 1, 1>': dict(launches=97, mfma_util=0.0, valu_busy=0.1)})
    monkeypatch.setattr(bench, 'REPO', str(tmp_path))
    os.makedirs(tmp_path / 'profiles' / 'r03')
    monkeypatch.setattr(bench, 'csrc_sha', lambda: doc['csrc_sha'])
    assert bench.pmc_mfma(key) is None                                                  # no file: nothing to quote
    path = tmp_path / 'profiles' / 'r03' / 'mfma_util.json'
    path.write_text(json.dumps(doc))
    m = bench.pmc_mfma(key)
    assert m['source'] == 'committed' and list(m['kernels']) == ['cell_bwd_x3_kernel<FmtH2, 32, 1, 0, 0>']      # the projection kernels only
    assert m['kernels']['cell_bwd_x3_kernel<FmtH2, 32, 1, 0, 0>'] == dict(launches=48, mfma_busy=0.3, valu_busy=0.4)
    path.write_text(json.dumps(dict(doc, csrc_sha='0' * 16)))
    assert bench.pmc_mfma(key)['value'] is None                                         # other kernel sources: never a stale number


def test_presets_and_global_batch():
    a = bench.parse(['--preset', 'cfg4'])
    assert (a.grid, a.order, a.batch_per_gpu) == (100, 3, 16)
    a = bench.parse(['--preset', 'cfg5'])
    assert (a.categories, a.storage) == (64, 'bf16')
    a = bench.parse(['--preset', 'sf'])
    assert (a.grid, a.categories, a.obs, a.pred, a.batch_per_gpu, a.graph_mode, a.hip_graph) == (10, 5, 9, 3, 32, 'csr-fixed', True)
    a = bench.parse(['--preset', 'sf-learned', '--eager'])            # an explicit flag beats the preset's default
    assert (a.graph_mode, a.hip_graph, a.no_cpu_baseline) == ('dense-learned', False, False)      # (the preset lines carry a CPU baseline since round 4)
    assert set(bench.PRICED_ENTRY_POINTS) >= set(bench.PLAIN_SPMM) | {'stc_cell_bwd_planar_f32'}
    a = bench.parse(['--gpus', '4', '--global-batch', '8'])
    assert a.batch_per_gpu == 2
    with pytest.raises(SystemExit):
        bench.parse(['--gpus', '4', '--global-batch', '6'])


def test_presets_block_runs_children_and_survives_their_failures(monkeypatch):
    """``run_presets``: one child process per BASELINE configuration (never a re-exec of the bench process), launcher variables stripped, the digest
    of each child's line; a child that fails, prints nothing or overruns leaves an ``error`` entry; nothing starts while the parent holds its memory."""
    import json
    import subprocess
    seen = []

    def fake_run(cmd, env=None, capture_output=None, text=None, timeout=None):
        seen.append((cmd, env))
        name = cmd[cmd.index('--preset') + 1]
        if name == 'cfg4':
            return subprocess.CompletedProcess(cmd, 1, stdout='', stderr='Traceback\nStcError: boom')
        if name == 'cfg5':
            raise subprocess.TimeoutExpired(cmd, timeout)
        if name == 'sf':
            return subprocess.CompletedProcess(cmd, 0, stdout='not json\n', stderr='')
        line = {'value': 7.0, 'unit': 'samples/s', 'ms_per_step': 3.0, 'steps': 3, 'hip_graph': True, 'dtype': 'f32', 'config': {'workload': 'w'},
                'roofline': {'bound': 'hbm', 'kernel': 'k', 'achieved': 1.0, 'peak': 2.0, 'unit': 'GB/s', 'frac': 0.5,
                             'dominant': {'entry_point': 'e', 'frac': 0.4, 'launches': 9}}}
        return subprocess.CompletedProcess(cmd, 0, stdout='warning\n' + json.dumps(line) + '\n', stderr='')

    monkeypatch.setattr(bench.subprocess, 'run', fake_run)
    monkeypatch.setenv('RANK', '0')
    monkeypatch.delenv('STC_BENCH_PRESETS', raising=False)
    d = bench.run_presets(0.5)
    assert [c[c.index('--preset') + 1] for c, _ in seen] == ['cfg2', 'cfg2', 'cfg4', 'cfg5', 'sf', 'sf-learned']
    assert all('RANK' not in e and c[0] == sys.executable and '--no-cpu-baseline' in c and c[c.index('--steps') + 1] == '3' for c, e in seen)
    assert d['cfg2']['value'] == 7.0 and d['cfg2']['roofline']['dominant'] == {'entry_point': 'e', 'frac': 0.4} and d['sf-learned']['roofline']['frac'] == 0.5
    assert 'boom' in d['cfg4']['error'] and 'no line within' in d['cfg5']['error'] and 'exit code 0' in d['sf']['error']
    seen.clear()
    assert 'error' in bench.run_presets(140.0) and not seen
