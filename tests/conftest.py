"""pytest configuration: GPU marker, import paths and golden-vector loading."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'stc-gnn_amd')
GOLDEN = os.path.join(REPO, 'tests', 'golden')
REFERENCE = '/root/reference/framework'

for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped, loudly, when no device is present and they were not deselected.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """Load tests/golden/<name>.npz as a dict of torch tensors (scalars stay numpy)."""
    out = {}
    with np.load(os.path.join(GOLDEN, name + '.npz')) as z:
        for k in z.files:
            a = z[k]
            out[k] = torch.from_numpy(a) if (a.ndim > 0 and a.dtype.kind in 'fiub') else (a if a.ndim > 0 else a[()])
    return out


def sub_dict(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def rel_err(a, b):
    """max |a-b| / max |b| : the relative error every parity test uses."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    denom = b.abs().max().item()
    return (a - b).abs().max().item() / (denom if denom > 0 else 1.0)


@pytest.fixture(scope='session')
def reference_module():
    """The live reference, only when /root/reference is present (never on the GPU box)."""
    if not os.path.isdir(REFERENCE):
        pytest.skip('reference not present')
    sys.dont_write_bytecode = True
    import importlib.util
    spec = importlib.util.spec_from_file_location('_ref_STC_GNN', os.path.join(REFERENCE, 'STC_GNN.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
