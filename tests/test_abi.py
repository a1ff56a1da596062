"""CPU: the C-ABI library loads and exports every symbol include/stc_hip.h declares; argument
validation that happens before any launch works without a GPU; the product fails loudly without one."""
import ctypes
import os
import re

import pytest
import torch

from stc_hip import _lib
from tests.conftest import REPO


def _declared_symbols():
    text = open(os.path.join(REPO, 'include', 'stc_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(stc_[a-z0-9_]+)\s*\(', text)))


def test_header_and_python_export_lists_agree():
    assert _declared_symbols() == sorted(_lib.EXPORTS)


def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), 'libstc_hip.so not built: python -c "import __graft_entry__ as g; g.build()"'
    lib = _lib.load_library()
    for name in _declared_symbols():
        assert hasattr(lib, name), f'{name} declared in include/stc_hip.h but not exported'
    assert lib.stc_version() == _lib.ABI_VERSION


def test_integration_doc_names_the_current_abi():
    """INTEGRATION.md is what a maintainer pastes from: its ABI number is the header's, and its sample reads the number from the header."""
    text = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    header = open(os.path.join(REPO, 'include', 'stc_hip.h')).read()
    abi = int(re.search(r'#define\s+STC_ABI_VERSION\s+(\d+)', header).group(1))
    assert abi == _lib.ABI_VERSION and f'ABI v{abi})' in text
    assert not re.search(r'stc_version\(\)\s*==\s*\d', text)


def test_argument_validation_needs_no_gpu():
    lib = _lib.load_library()
    # negative sizes / null pointers are rejected before any HIP call
    rc = lib.stc_csr_spmm_f32(None, None, None, -1, 4, None, None, None, 1, 8, 1.0, 0.0, None)
    assert rc == -1 and b'negative' in lib.stc_last_error()
    rc = lib.stc_csr_spmm_f32(None, None, None, 4, 4, None, None, None, 1, 8, 1.0, 0.0, None)
    assert rc == -1 and b'null' in lib.stc_last_error()
    assert lib.stc_csr_spmm_f32(None, None, None, 0, 4, None, None, None, 1, 8, 1.0, 0.0, None) == 0   # empty: no launch
    arr = (ctypes.c_void_p * 1)()
    rc = lib.stc_bdg_node_fwd_f32(arr, 9, None, 2, None, None, None, 10, 3, 5, 5, 4, None)
    assert rc == -3 and b'Chebyshev' in lib.stc_last_error()          # order above STC_MAX_K
    assert lib.stc_bdg_node_bwd_workspace_bytes(2, 2, 32, 32, 32, 1) == 512 * 4 * (2 * 2 * 32 * 32 + 32 + 2 * 32 * 32)
    assert lib.stc_gru_blend_fwd_f32(None, None, None, None, None, 0, None) == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason='only meaningful without a GPU')
def test_product_fails_loudly_without_gpu():
    """No CPU fallback: the kernel front refuses to come up, and the drop-in module raises on CPU tensors."""
    import STC_GNN as M
    from stc_hip import ops
    assert ops._kernels is None or ops._kernels.name == 'hip-gfx950'
    with pytest.raises(_lib.StcError):
        _lib.HipKernels()
    layer = M.BDG_Dif(2, 2, 5, 4)
    with pytest.raises(_lib.StcError):
        layer(torch.randn(2, 12, 3, 5), torch.randn(12, 12), torch.randn(3, 3))


def test_missing_library_message(tmp_path, monkeypatch):
    monkeypatch.setattr(_lib, '_LIB', None)
    with pytest.raises(_lib.StcError, match='not built'):
        _lib.load_library(str(tmp_path / 'nope.so'))


def test_combo_loss_matches_oracle():
    from oracle import stc_oracle as O
    from stc_hip.loss import ComboLoss
    g = torch.Generator().manual_seed(0)
    p = torch.rand(3, 2, 12, 3, generator=g).clamp(0.01, 0.99).requires_grad_()
    y = (torch.rand(3, 2, 12, 3, generator=g) < 0.3).float()
    a = ComboLoss()(p, y)
    b = O.combo_loss(p.detach().clone().requires_grad_(), y)
    assert abs(float(a.detach()) - float(b.detach())) < 1e-7


def test_row_block_plan_reconstructs_the_graph():
    """Host logic of the row-blocked SpMM: the BCSR arrays must hold exactly the entries of the CSR matrix."""
    import numpy as np
    from stc_hip import CsrGraph
    from stc_hip.graph import BLOCK_BATCH, BLOCK_ROWS
    for g in (CsrGraph.queen_grid(13, 9), CsrGraph.queen_grid(7, 5, permute_seed=3), CsrGraph.from_dense(torch.zeros(6, 6))):
        dense = g.to_dense().numpy()
        for side, want in (('fwd', dense.T), ('bwd', dense)):
            h = g._host
            bp, bc, bv = h[f'{side}_blk_ptr'], h[f'{side}_blk_cols'], h[f'{side}_blk_vals']
            got = np.zeros_like(want)
            assert np.all(np.diff(bp) % BLOCK_BATCH == 0)              # whole gather batches: no remainder loop in the kernel
            for blk in range(len(bp) - 1):
                cols, vals = bc[bp[blk]:bp[blk + 1]], bv[bp[blk]:bp[blk + 1]]
                real = np.concatenate([[True], np.diff(cols) > 0]) if cols.size else np.zeros(0, bool)
                assert np.all(np.diff(cols) >= 0) and real[:int(real.sum())].all()   # sorted, distinct, then the padding
                assert not vals[~real].any() and np.all(cols[~real] == cols[real][-1] if (~real).any() else True)   # zero-weight repeats of the last column
                assert (~real).sum() < BLOCK_BATCH
                for r in range(BLOCK_ROWS):
                    if blk * BLOCK_ROWS + r < g.n:
                        got[blk * BLOCK_ROWS + r, cols[real]] = vals[real, r]
                    else:
                        assert not vals[:, r].any()                    # rows past the end carry zeros
            assert np.array_equal(got, want)
            assert bp[-1] == bc.size == bv.shape[0] and bv.shape[1] == BLOCK_ROWS


def test_ctypes_signatures_match_the_header():
    """Every prototype in include/stc_hip.h has the same number and kinds of parameters as its ctypes argtypes
    (a mismatch only shows up as a TypeError at the first launch on the GPU box otherwise)."""
    import ctypes as C
    text = open(os.path.join(REPO, 'include', 'stc_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    lib = _lib.load_library()
    protos = re.findall(r'\b(?:int|size_t|const char\*)\s+(stc_[a-z0-9_]+)\s*\(([^)]*)\)\s*;', text)
    assert len(protos) == len(_lib.EXPORTS)
    kind = {C.c_void_p: 'ptr', C.c_int32: 'i32', C.c_int64: 'i64', C.c_float: 'f32', C.c_double: 'f64', C.c_size_t: 'size', C.POINTER(C.c_void_p): 'ptr',
            C.POINTER(C.c_int32): 'ptr', C.POINTER(C.c_float): 'ptr'}
    for name, params in protos:
        want = []
        for prm in [q.strip() for q in params.split(',') if q.strip() and q.strip() != 'void']:
            if '*' in prm:
                want.append('ptr')
            elif prm.startswith('int32_t'):
                want.append('i32')
            elif prm.startswith('int64_t'):
                want.append('i64')
            elif prm.startswith('float'):
                want.append('f32')
            elif prm.startswith('double'):
                want.append('f64')
            elif prm.startswith('size_t'):
                want.append('size')
            else:
                raise AssertionError(f'{name}: unparsed parameter {prm!r}')
        got = [kind[t] for t in (getattr(lib, name).argtypes or [])]
        assert got == want, f'{name}: ctypes {got} vs header {want}'
