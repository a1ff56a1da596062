"""ctypes binding of ``libstc_hip.so`` (C ABI: ``include/stc_hip.h``).

``HipKernels`` exposes one method per exported kernel, taking torch tensors
that live on the MI355X: it checks device / dtype / contiguity / shapes on
the host (a wrong shape handed to a hand-written kernel can fault the GPU),
passes raw device pointers plus the current HIP stream, and turns a non-zero
return code into ``StcError`` carrying ``stc_last_error()``.

There is no CPU implementation behind this class: if the shared library is
missing or a tensor is not a ROCm tensor the call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from .graph import is_full_pattern

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG_ROOT, 'libstc_hip.so')
ABI_VERSION = 33
FMT_BF16X3, FMT_F16X2 = 0, 1          # STC_FMT_* of include/stc_hip.h: operand formats of the split-operand matrix-core kernels
MAX_K = 4
SPMM_SUM_MAX_ADD = 8     # = STC_SPMM_SUM_MAX_ADD of include/stc_hip.h
PATCH_ROWS, PATCH_MAX_SRC = 32, 64          # = STC_PATCH_ROWS, STC_PATCH_MAX_SRC

#: every symbol ``include/stc_hip.h`` declares (the CPU test-suite checks the .so exports them all)
EXPORTS = (
    'stc_version', 'stc_last_error',
    'stc_csr_spmm_f32', 'stc_bcsr_spmm_f32', 'stc_patch_spmm_f32', 'stc_patch_spmm_bf16', 'stc_ring2_sum_f32', 'stc_ring2_blend_f32', 'stc_ring2_chain_f32', 'stc_csr_spmm_bf16', 'stc_bcsr_spmm_bf16', 'stc_bdg_node_bf16_supported', 'stc_bdg_node_fwd_bf16', 'stc_bdg_node_bwd_bf16',
    'stc_cell_planar_bf16_supported', 'stc_cell_gates_fwd_planar_bf16', 'stc_cell_gates_bwd_planar_bf16', 'stc_bdg_node_post_bwd_bf16',
    'stc_cell_bwd_planar_bf16_supported', 'stc_cell_bwd_planar_bf16',
    'stc_spmm_blend_fwd_bf16', 'stc_spmm_sum_bf16', 'stc_gru_blend_bwd_bf16', 'stc_head_fwd_bf16', 'stc_head_bwd_bf16',
    'stc_csr_sddmm_f32', 'stc_set_dispatch_level', 'stc_dense_agg_f32',
    'stc_cheby_dense_fwd_f32', 'stc_cheby_dense_bwd_f32',
    'stc_bdg_node_fwd_f32', 'stc_bdg_node_bwd_workspace_bytes', 'stc_bdg_node_bwd_f32',
    'stc_bdg_node_post_supported', 'stc_bdg_node_post_fwd_f32', 'stc_bdg_node_post_bwd_f32', 'stc_spmm_blend_fwd_f32',
    'stc_cell_fused_supported', 'stc_cell_gates_fwd_f32', 'stc_cell_gates_bwd_f32', 'stc_cell_cand_bwd_f32', 'stc_cell_blend_fwd_f32',
    'stc_cell_planar_supported', 'stc_cell_gates_fwd_planar_f32', 'stc_cell_gates_bwd_planar_f32', 'stc_spmm_sum_f32',
    'stc_cell_bwd_planar_supported', 'stc_cell_bwd_planar_workspace_bytes', 'stc_cell_bwd_planar_f32',
    'stc_cell_planar_k_supported', 'stc_cell_gates_fwd_planar_k_f32', 'stc_cell_cand_fwd_planar_k_f32', 'stc_cell_gates_bwd_planar_k_f32',
    'stc_cell_cand_bwd_planar_k_f32',
    'stc_cell_small_supported', 'stc_cell_small_workspace_bytes', 'stc_cell_small_param_rows', 'stc_graph_grad_f32', 'stc_mix_grad_f32', 'stc_mix_dt_supported', 'stc_mix_dt_workspace_bytes', 'stc_mix_dt_f32', 'stc_mixed_fusion_workspace_bytes', 'stc_mixed_fusion_fwd_f32', 'stc_mixed_fusion_bwd_f32', 'stc_mgp_uv_fwd_f32', 'stc_mgp_uv_bwd_f32', 'stc_mgp_softmax_fwd_f32', 'stc_mgp_softmax_bwd_f32', 'stc_adam_f32', 'stc_cell_small_fwd_f32', 'stc_cell_small_bwd_f32',
    'stc_gru_gates_fwd_f32', 'stc_gru_gates_bwd_f32', 'stc_gru_blend_fwd_f32', 'stc_gru_blend_bwd_f32',
    'stc_head_fwd_f32', 'stc_head_bwd_workspace_bytes', 'stc_head_bwd_f32',
    'stc_axpy_f32', 'stc_concat2_f32', 'stc_split2_f32',
)


class StcError(RuntimeError):
    """A C-ABI call returned non-zero, or the HIP extension cannot be used."""


_p = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f32 = C.c_float


def _declare(lib):
    lib.stc_version.restype = C.c_int
    lib.stc_version.argtypes = []
    lib.stc_last_error.restype = C.c_char_p
    lib.stc_last_error.argtypes = []
    sig = {
        'stc_csr_spmm_f32': [_p, _p, _p, _i32, _i32, _p, _p, _p, _i32, _i32, _f32, _f32, _p],
        'stc_bcsr_spmm_f32': [_p, _p, _p, _i32, _i32, _p, _p, _p, _i32, _i32, _f32, _f32, _p],
        'stc_patch_spmm_f32': [_p] * 5 + [_i32] * 4 + [_p, _p, _p, _i32, _i32, _f32, _f32, _p],
        'stc_patch_spmm_bf16': [_p] * 5 + [_i32] * 4 + [_p, _p, _p, _i32, _i32, _f32, _f32, _p],
        'stc_csr_spmm_bf16': [_p, _p, _p, _i32, _i32, _p, _p, _p, _i32, _i32, _f32, _f32, _p],
        'stc_bcsr_spmm_bf16': [_p, _p, _p, _i32, _i32, _p, _p, _p, _i32, _i32, _f32, _f32, _p],
        'stc_dense_agg_f32': [_p, _i32, _i32, _p, _p, _p, _i32, _i32, _f32, _f32, _p],
        'stc_cell_small_fwd_f32': [_p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _i32, _p, _p, _i32, _i32] + [_p] * 16 + [_i32, _i32, _i32, _i32, _p],
        'stc_graph_grad_f32': [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _p],
        'stc_mix_grad_f32': [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _p],
        'stc_mixed_fusion_workspace_bytes': [_i32, _i32],
        'stc_mixed_fusion_fwd_f32': [_p] * 8 + [_i32, _p],
        'stc_mixed_fusion_bwd_f32': [_p] * 12 + [C.c_size_t, _i32, _p],
        'stc_mgp_uv_fwd_f32': [_p, _i64, _i64, _i64, _p, _p, _f32, _p, _p, _i32, _i32, _i32, _i32, _p],
        'stc_mgp_uv_bwd_f32': [_p, _i64, _i64, _i64, _p, _p, _p, _p, _f32, _p, _i32, _i32, _i32, _i32, _p],
        'stc_mgp_softmax_fwd_f32': [_p, _p, _i32, _p],
        'stc_mgp_softmax_bwd_f32': [_p, _p, _p, _p, _p, _i32, _p],
        'stc_adam_f32': [_p] * 4 + [_i64, _p] + [C.c_double] * 5 + [_p],
        'stc_cell_small_bwd_f32': [_p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _i32, _p, _p, _i32, _i32] + [_p] * 12 + [_i32, _p, _i32, _p, _i64, _i32, _i32,
                                   _p, _p, _p, _p, _p, C.c_size_t, _i32, _i32, _i32, _i32, _p],
        'stc_csr_sddmm_f32': [_p, _p, _i32, _i32, _p, _p, _p, _i32, _i32, _f32, _i32, _p],
        'stc_cheby_dense_fwd_f32': [_p, _i32, _i32, _p, _p],
        'stc_mix_dt_f32': [C.POINTER(_p), _i32, _p, _p, _p, _p, C.c_size_t, _i64, _i32, _i32, _i32, _i32, _p],
        'stc_cheby_dense_bwd_f32': [_p, _p, _p, _i32, _i32, _p, _p],
        'stc_bdg_node_fwd_f32': [C.POINTER(_p), _i32, _p, _i32, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p],
        'stc_bdg_node_bwd_f32': [C.POINTER(_p), _i32, _p, _i32, _p, _p, C.POINTER(_p), _p, _p, _p,
                                 _p, C.c_size_t, _i64, _i32, _i32, _i32, _i32, _p],
        'stc_bdg_node_fwd_bf16': [C.POINTER(_p), _i32, _p, _i32, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p],
        'stc_bdg_node_bwd_bf16': [C.POINTER(_p), _i32, _p, _i32, _p, _p, C.POINTER(_p), _p, _p,
                                  _p, C.c_size_t, _i64, _i32, _i32, _i32, _i32, _p],
        'stc_cell_gates_fwd_planar_bf16': [_p] * 14 + [_i64, _i32, _i32, _i32, _p],
        'stc_cell_gates_bwd_planar_bf16': [_p] * 11 + [C.POINTER(_p), _p, _p, _p, _p, C.c_size_t, _i64, _i32, _i32, _i32, _p],
        'stc_cell_bwd_planar_bf16': [_p] * 20 + [_p, C.c_size_t, _i64, _i32, _i32, _i32, _p],
        'stc_ring2_sum_f32': [_p] * 5 + [_i32, _i32, _p, _p, _i32, C.POINTER(_p), _p, _p, _p, _p, _i32, _i32, _i32, _p],
        'stc_ring2_blend_f32': [_p] * 5 + [_i32, _i32] + [_p] * 7 + [_i32, _i32, _i32, _p],
        'stc_ring2_chain_f32': [_p] * 5 + [_i32, _i32, _p, _p, _f32, _i32, C.POINTER(_p), _p, _f32, _i32, C.POINTER(_p), C.POINTER(_f32), _p, _i32, _i32, _i32, _p],
        'stc_bdg_node_post_bwd_bf16': [_p] * 11 + [C.c_size_t, _i64, _i32, _i32, _i32, _p],
        'stc_spmm_blend_fwd_bf16': [_p] * 6 + [_i32, _i32] + [_p] * 6 + [_i32] * 3 + [_p],
        'stc_spmm_sum_bf16': [_p] * 6 + [_i32, _i32, _p, _p, _i32, C.POINTER(_p), _p, _p, _p, _p, _i32, _i32, _i32, _p],
        'stc_gru_blend_bwd_bf16': [_p, _p, _p, _p, _i64, _p],
        'stc_head_fwd_bf16': [_p, _p, _p, _p, _i64, _i32, _p],
        'stc_head_bwd_bf16': [_p, _p, _p, _p, _p, _p, _p, C.c_size_t, _i64, _i32, _p],
        'stc_bdg_node_post_fwd_f32': [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p],
        'stc_cell_gates_fwd_planar_f32': [_p] * 14 + [_i32, _p, _i64, _i32, _i32, _i32, _p],
        'stc_cell_gates_bwd_planar_f32': [_p] * 11 + [C.POINTER(_p), _p, _p, _p, _i32, _p, _p, C.c_size_t, _i64, _i32, _i32, _i32, _p],
        'stc_cell_bwd_planar_f32': [_p] * 20 + [_i32, _i32, _i32, _p, _p, C.c_size_t, _i64, _i32, _i32, _i32, _p],
        'stc_spmm_blend_fwd_f32': [_p] * 6 + [_i32, _i32] + [_p] * 6 + [_p, _i32, _i32, _p, _i32, _p, _i32, _i32] + [_i32] * 3 + [_p],
        'stc_bdg_node_post_bwd_f32': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p, C.c_size_t, _i64, _i32, _i32, _i32, _i32, _p],
        'stc_spmm_sum_f32': [_p] * 6 + [_i32, _i32, _p, _p, _f32, _i32, C.POINTER(_p), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_f32), _p, _p, _p, _p,
                             _p, _i32, _i32, _i32, _i32, _p],
        'stc_cell_gates_fwd_planar_k_f32': [C.POINTER(_p), C.POINTER(_p), _i32, _p, _p, _p, _p, _p, _p, _i32, _p, _i64, _i32, _i32, _i32, _p],
        'stc_cell_cand_fwd_planar_k_f32': [C.POINTER(_p), C.POINTER(_p), _i32, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _i64, _i32, _i32, _i32, _p],
        'stc_cell_gates_bwd_planar_k_f32': [C.POINTER(_p), C.POINTER(_p), _i32, _p, _p, _p, _p, _p, _p, _p, C.POINTER(_p), C.POINTER(_p), _p, _p, _p, _i32,
                                            _i32, _p, _p, C.c_size_t, _i64, _i32, _i32, _i32, _p],
        'stc_cell_cand_bwd_planar_k_f32': [C.POINTER(_p), C.POINTER(_p), _i32, _p, _p, _p, _p, _p, C.POINTER(_p), C.POINTER(_p), _p, _p,
                                           _i32, _p, _p, C.c_size_t, _i64, _i32, _i32, _i32, _p],
        'stc_cell_gates_fwd_f32': [C.POINTER(_p), _i32, _p, _i32, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _p],
        'stc_cell_cand_bwd_f32': [C.POINTER(_p), _i32, _p, _i32, _p, _p, _p, _p, C.POINTER(_p), _p, _p, _p, C.c_size_t, _i64, _i32, _i32, _i32, _i32, _p],
        'stc_cell_gates_bwd_f32': [C.POINTER(_p), _i32, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _i32, C.POINTER(_p), _p, _p, _p, _p,
                                   _p, C.c_size_t, _i64, _i32, _i32, _i32, _i32, _i32, _p],
        'stc_cell_blend_fwd_f32': [C.POINTER(_p), _i32, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _p, _i32, _p, _i32, _i32,
                                   _i64, _i32, _i32, _i32, _i32, _p],
        'stc_gru_gates_fwd_f32': [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p],
        'stc_gru_gates_bwd_f32': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p],
        'stc_gru_blend_fwd_f32': [_p, _p, _p, _p, _p, _i64, _p],
        'stc_gru_blend_bwd_f32': [_p, _p, _p, _p, _p, _p, _p, _i64, _p],
        'stc_head_fwd_f32': [_p, _p, _p, _p, _i64, _i32, _p],
        'stc_head_bwd_f32': [_p, _p, _p, _p, _p, _p, _p, C.c_size_t, _i64, _i32, _p],
        'stc_axpy_f32': [_f32, _p, _p, _i64, _p],
        'stc_concat2_f32': [_p, _p, _p, _i64, _i32, _i32, _i32, _p],
        'stc_split2_f32': [_p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = argtypes
    lib.stc_cell_fused_supported.restype = C.c_int
    lib.stc_cell_fused_supported.argtypes = [_i32, _i32, _i32, _i32, _i32]
    lib.stc_bdg_node_post_supported.restype = C.c_int
    lib.stc_bdg_node_post_supported.argtypes = [_i32, _i32, _i32, _i32, _i32]
    lib.stc_cell_planar_bf16_supported.restype = C.c_int
    lib.stc_cell_planar_bf16_supported.argtypes = [_i32, _i32, _i32, _i32]
    lib.stc_bdg_node_bf16_supported.restype = C.c_int
    lib.stc_bdg_node_bf16_supported.argtypes = [_i32, _i32, _i32, _i32, _i32]
    lib.stc_cell_planar_supported.restype = C.c_int
    lib.stc_cell_planar_supported.argtypes = [_i32, _i32, _i32, _i32]
    lib.stc_cell_planar_k_supported.restype = C.c_int
    lib.stc_cell_planar_k_supported.argtypes = [_i32, _i32, _i32]
    lib.stc_cell_bwd_planar_supported.restype = C.c_int
    lib.stc_cell_bwd_planar_supported.argtypes = [_i32, _i32]
    lib.stc_cell_bwd_planar_bf16_supported.restype = C.c_int
    lib.stc_cell_bwd_planar_bf16_supported.argtypes = [_i32, _i32, _i32]
    lib.stc_cell_bwd_planar_workspace_bytes.restype = C.c_size_t
    lib.stc_cell_bwd_planar_workspace_bytes.argtypes = [_i32, _i32, _i32]
    lib.stc_cell_small_supported.restype = C.c_int
    lib.stc_cell_small_supported.argtypes = [_i32, _i32, _i32, _i32, _i32]
    lib.stc_cell_small_param_rows.restype = C.c_int
    lib.stc_cell_small_param_rows.argtypes = []
    lib.stc_cell_small_workspace_bytes.restype = C.c_size_t
    lib.stc_cell_small_workspace_bytes.argtypes = [_i32, _i32, _i32, _i32, _i32]
    lib.stc_set_dispatch_level.restype = C.c_int
    lib.stc_set_dispatch_level.argtypes = [_i32]
    lib.stc_head_bwd_workspace_bytes.restype = C.c_size_t
    lib.stc_mixed_fusion_workspace_bytes.argtypes = [_i32, _i32]
    lib.stc_mixed_fusion_workspace_bytes.restype = C.c_size_t
    lib.stc_head_bwd_workspace_bytes.argtypes = [_i32]
    lib.stc_bdg_node_bwd_workspace_bytes.restype = C.c_size_t
    lib.stc_bdg_node_bwd_workspace_bytes.argtypes = [_i32, _i32, _i32, _i32, _i32, _i32]
    lib.stc_mix_dt_supported.restype = C.c_int
    lib.stc_mix_dt_supported.argtypes = [_i32] * 5
    lib.stc_mix_dt_workspace_bytes.restype = C.c_size_t
    lib.stc_mix_dt_workspace_bytes.argtypes = [_i32]


_LIB = None


def load_library(path: str = LIB_PATH):
    """dlopen the kernel library (once) and check its ABI version."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(path):
        raise StcError(
            f'{path} not found: the HIP kernels are not built. Run '
            f'`make -C {os.path.join(_PKG_ROOT, "csrc")}` (or `python -c "import __graft_entry__ as g; g.build()"`). '
            'There is no CPU fallback for the STC-GNN hot path.')
    lib = C.CDLL(path)
    _declare(lib)
    v = lib.stc_version()
    if v != ABI_VERSION:
        raise StcError(f'{path}: ABI version {v}, host expects {ABI_VERSION}')
    _LIB = lib
    return lib


class KernelTimer:
    """Per-launch HIP-event timing for ``bench.py``: events are recorded on the stream the kernel is
    launched on (torch's current stream), immediately around the launch."""

    def __init__(self, only=None):
        self.spans = []          # (name, start_event, end_event, algorithmic_bytes, tag)
        self.only = None if only is None else frozenset(only)      # time these entry points only (the others launch without event records)

    def add(self, name, start, end, nbytes, tag=None):
        self.spans.append((name, start, end, nbytes, tag))

    def summary(self):
        """{name: dict(launches, ms, bytes[, tags: {tag: dict(launches, ms, bytes)}])} after synchronising; clears the
        recorded spans.  A tag names the form of a launch where one entry point has several ('plain' = Y = S.X with no Y0)."""
        torch.cuda.synchronize()
        out = {}
        for name, s, e, nb, tag in self.spans:
            d = out.setdefault(name, dict(launches=0, ms=0.0, bytes=0))
            ms = s.elapsed_time(e)
            for acc in (d,) if tag is None else (d, d.setdefault('tags', {}).setdefault(tag, dict(launches=0, ms=0.0, bytes=0))):
                acc['launches'] += 1
                acc['ms'] += ms
                acc['bytes'] += nb
        self.spans = []
        return out


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class HipKernels:
    """Tensor-level front of the C ABI; the one object ``stc_hip.ops`` launches through."""

    name = 'hip-gfx950'
    #: the planar gates backward adds the state's share from its gate prologue into the H plane's gradient itself (dH=None)
    folds_dH = True
    #: operand format of the split-operand matrix-core cell kernels (include/stc_hip.h "operand formats"): two fp16 pieces / three
    #: products by default, STC_OPERAND_FORMAT=bf16x3 keeps three bf16 pieces / six products (fp32's range, twice the matrix instructions)
    operand_format = {'f16x2': FMT_F16X2, 'bf16x3': FMT_BF16X3}[os.environ.get('STC_OPERAND_FORMAT', 'f16x2')]
    #: Graphs whose largest absolute row sum (either orientation; to the power K - 1 for Chebyshev order K) exceeds this run the planar cell kernels on the 24-bit format (bf16 x 3) even
    #: when fp16 x 2 is the default: an aggregation amplifies a state -- and its rounding noise -- by up to that factor per cell step, and
    #: where the model amplifies noise the 22-bit operands show as 4-5x the reference's own fp32 noise (row sums of 50: 2.7e-5 on the prediction
    #: against a reference noise of 5e-6; bf16 x 3: 7-10e-6; row sums of 16: both 1e-6 -- tests/test_scale_sweep.py).  Row-stochastic graphs
    #: (the bench's, the reference's softmax part) have 1 .. 1.5, the reference's raw 0/1 adjacency 8.
    HEAVY_ROW_SUM = 24.0
    #: the patch form of the plain aggregation where the graph has one (STC_PATCH_SPMM=0: always the row-blocked kernel -- for A/B timing)
    patch_spmm = os.environ.get('STC_PATCH_SPMM', '1') != '0'
    #: ... for launches of at least this many (patch, sample) workgroups with a Y0 operand (three / six times as many without, see csr_spmm)
    patch_min_items = 1000

    def __init__(self):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise StcError('no ROCm device visible: the STC-GNN hot path runs on MI355X only (no CPU fallback)')
        self._workspace = {}
        self._retired = []                            # outgrown workspaces, kept alive (see _get_workspace)
        self.timer: Optional[KernelTimer] = None      # set by bench.py to time every launch with HIP events

    ACT_AMAX_SLOTS = 256         # STC_ACT_AMAX_SLOTS: floats per plane row of an activation-maximum buffer

    def act_amax_buffer(self, like, *lead):
        """Zero-filled slots for the plane maxima of fp16 x 2 forward launches, shape (*lead, 256): a forward launch fills ``planes`` rows
        (``act_amax=``) and the matching backward launch reads them; None on the bf16 x 3 format (no range information needed)."""
        if self.operand_format != FMT_F16X2:
            return None
        return torch.zeros(*lead, self.ACT_AMAX_SLOTS, dtype=torch.float32, device=like.device)

    def _act_amax(self, what, given, rows, on):
        """Pointer of a launch's activation-maximum slots (``rows`` plane rows) or None: the fp16 x 2 format only."""
        if given is None or self.operand_format != FMT_F16X2:
            return None
        self._f32(what + '.act_amax', given, (rows, self.ACT_AMAX_SLOTS))
        self._same_device(on, given)
        return given.data_ptr()

    def for_graph(self, row_sum_bound: float):
        """The kernel set a schedule on a graph with that row-sum bound launches through: ``self``, or -- fp16 x 2 default, heavy graph -- a
        view of it on the bf16 x 3 format (same library, workspaces and timer)."""
        if self.operand_format != FMT_F16X2 or not (row_sum_bound > self.HEAVY_ROW_SUM):
            return self
        view = getattr(self, '_b3_view', None)
        if view is None:
            import copy
            view = copy.copy(self)
            view.operand_format = FMT_BF16X3
            view._bf16_front = None
            self._b3_view = view
        view.timer = self.timer
        return view

    def set_dispatch_level(self, level: int):
        """0 = every kernel path (default), 1 = no split-operand matrix-core kernels (fp32 MFMA instead), 2 = generic kernels only:
        process-wide ceiling of the node / cell kernel dispatch (stc_set_dispatch_level), for tests and A/B runs."""
        rc = self.lib.stc_set_dispatch_level(int(level))
        if rc != 0:
            raise StcError(f'stc_set_dispatch_level({level}) failed: {self.lib.stc_last_error().decode()}')

    def _launch(self, name, on, *args, nbytes=0, tag=None):
        """Call C entry point ``name`` with ``args`` + the current stream of ``on``'s device.

        The host side of a launch matters at small shapes: at the SF shape a train step is ~700 launches of ~7 us of GPU time each, and the
        ``torch.cuda.device`` context + ``torch.cuda.current_stream`` objects alone cost more than that per launch
        (tools/probes/sf_host_profile.py).  So: the raw stream handle straight from torch's C layer, and a device switch only when the
        tensor does not live on the current device."""
        fn = getattr(self.lib, name)
        index = on.device.index
        timer = self.timer
        if timer is not None and timer.only is not None and name not in timer.only:
            timer = None
        if timer is None and index == torch._C._cuda_getDevice():
            rc = fn(*args, torch._C._cuda_getCurrentRawStream(index))
        else:
            with torch.cuda.device(on.device):
                stream = torch.cuda.current_stream(on.device)
                if timer is None:
                    rc = fn(*args, stream.cuda_stream)
                else:
                    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    start.record(stream)
                    rc = fn(*args, stream.cuda_stream)
                    end.record(stream)
                    timer.add(name, start, end, nbytes, tag)
        if rc != 0:
            msg = self.lib.stc_last_error()
            raise StcError(f'{name} failed with code {rc}: {msg.decode() if msg else "?"}')

    # ---- host-side checks -------------------------------------------------------
    @staticmethod
    def _f32(name, t, shape=None):
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise StcError(f'{name}: expected a ROCm (cuda) tensor, got {type(t).__name__}'
                           f'{"" if not isinstance(t, torch.Tensor) else " on " + str(t.device)}; no CPU fallback')
        if t.dtype != torch.float32:
            raise StcError(f'{name}: expected float32, got {t.dtype} (the reference path is fp32-only)')
        if not t.is_contiguous():
            raise StcError(f'{name}: tensor must be contiguous')
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise StcError(f'{name}: shape {tuple(t.shape)}, expected {tuple(shape)}')
        return t

    @staticmethod
    def _bf16(name, t, shape=None):
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise StcError(f'{name}: expected a ROCm (cuda) tensor; no CPU fallback')
        if t.dtype != torch.bfloat16:
            raise StcError(f'{name}: expected bfloat16, got {t.dtype}')
        if not t.is_contiguous():
            raise StcError(f'{name}: tensor must be contiguous')
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise StcError(f'{name}: shape {tuple(t.shape)}, expected {tuple(shape)}')
        return t

    @staticmethod
    def _i32(name, t, numel=None):
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous():
            raise StcError(f'{name}: expected a contiguous int32 ROCm tensor')
        if numel is not None and t.numel() != numel:
            raise StcError(f'{name}: {t.numel()} elements, expected {numel}')
        return t

    @staticmethod
    def _stream(t):
        return torch.cuda.current_stream(t.device).cuda_stream

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.stc_last_error()
            raise StcError(f'{what} failed with code {rc}: {msg.decode() if msg else "?"}')

    def _same_device(self, *ts):
        devs = {t.device for t in ts if t is not None}
        if len(devs) > 1:
            raise StcError(f'tensors on different devices: {devs}')

    # ---- spatial aggregation ------------------------------------------------------
    def csr_spmm(self, rowptr, colidx, val, n_rows, n_cols, X, Y0, Y, alpha, beta, plan=None):
        """``plan`` = (blk_ptr, blk_cols, blk_vals) of ``graph._row_block_plan`` (fixed graphs): use the
        row-blocked kernel when the operands allow it (F % 4 == 0, 16-byte aligned); otherwise, or without a
        plan (learned dense graph: values change every step), the CSR kernel."""
        if X.dtype == torch.bfloat16:                       # bf16 rows: the bf16-storage kernels (fp32 values and sums)
            return self.csr_spmm_bf16(rowptr, colidx, val, n_rows, n_cols, X, Y0, Y, alpha, beta, plan=plan)
        B, nc, F = X.shape
        self._f32('spmm.X', X, (B, n_cols, F))
        self._f32('spmm.Y', Y, (B, n_rows, F))
        if Y0 is not None:
            self._f32('spmm.Y0', Y0, (B, n_rows, F))
        self._i32('spmm.rowptr', rowptr, n_rows + 1)
        self._i32('spmm.colidx', colidx)
        self._f32('spmm.val', val, (colidx.numel(),))
        self._same_device(rowptr, colidx, val, X, Y0, Y)
        plain = Y0 is None or beta == 0
        if plan is None:
            from .graph import is_full_pattern
            if is_full_pattern(colidx, n_rows, n_cols):       # a learned dense graph: the product is dense -> exact-fp32 matrix cores
                self._launch('stc_dense_agg_f32', X, _ptr(val), n_rows, n_cols, _ptr(X), _ptr(Y0), _ptr(Y), B, F, float(alpha), float(beta),
                             nbytes=4 * n_rows * n_cols + (2 if plain else 3) * 4 * B * n_rows * F, tag='dense')
                return
        nbytes = colidx.numel() * 8 + 4 * (n_rows + 1) + (2 if plain else 3) * 4 * B * n_rows * F
        tag = 'plain' if plain else 'with_y0'
        # row-blocked kernel: rows of >= 64 floats, or narrow rows of 4 / 8 / 16 / 32 floats (the layer-0 input plane: several row
        # blocks per wave); anything else (odd widths, unaligned operands, no plan) goes to the CSR kernels
        aligned = all(t is None or t.data_ptr() % 16 == 0 for t in (X, Y0, Y))
        # patch form (a graph whose rows cluster, rows in whole 1 KiB chunks): source rows staged through LDS, copy rate
        # (one workgroup per (patch, sample).  With few rounds of the chip's 512 resident workgroups the tail of the launch costs more than
        #  the staging saves -- rows of 512 floats, Y = S.X on the bench's grid: 37.7 against 32.7 us for one sample (1 594 workgroups), 83
        #  against 88 us for two; on a 100 x 100 grid with four samples (1 272): 29.6 against 28.0 us; rows of 1 024 floats (four chunks per
        #  workgroup; splitting them over two workgroups was slower still): 96 against 86 us for one sample, 169 against 165 for two.  With a
        #  Y0 operand the patch form wins in all of these (54.6 against 61.6, 108 against 128, 41.5 against 49.7, 127 against 132 us).
        #  Hence: from 1 000 workgroups with Y0, 3 000 without for rows of <= 512 floats, 6 000 for wider rows.)
        if F % 256 == 0 and aligned and self._patch_wanted(plan, plan[3][3].shape[0] * B if plan is not None and len(plan) > 3 else 0, F > 512, plain):
            self._launch('stc_patch_spmm_f32', X, *self._patch_ptrs(plan[3], X),
                         n_rows, n_cols, _ptr(X), _ptr(Y0), _ptr(Y), B, F, float(alpha), float(beta), nbytes=nbytes, tag=tag)
            return
        if plan is not None and F % 4 == 0 and (F >= 64 or F in (4, 8, 16, 32)) and aligned:
            blk_ptr, blk_cols, blk_vals = plan[:3]
            self._i32('spmm.blk_ptr', blk_ptr, (n_rows + 3) // 4 + 1)
            self._i32('spmm.blk_cols', blk_cols)
            self._f32('spmm.blk_vals', blk_vals, (blk_cols.numel(), 4))
            self._launch('stc_bcsr_spmm_f32', X, _ptr(blk_ptr), _ptr(blk_cols), _ptr(blk_vals), n_rows, n_cols,
                         _ptr(X), _ptr(Y0), _ptr(Y), B, F, float(alpha), float(beta), nbytes=nbytes, tag=tag if F >= 64 else tag + '_narrow_rows')
            return
        self._launch('stc_csr_spmm_f32', X, _ptr(rowptr), _ptr(colidx), _ptr(val), n_rows, n_cols, _ptr(X), _ptr(Y0), _ptr(Y), B, F, float(alpha), float(beta),
                     nbytes=nbytes, tag=tag)

    def csr_spmm_bf16(self, rowptr, colidx, val, n_rows, n_cols, X, Y0, Y, alpha, beta, plan=None):
        """bf16-storage form of ``csr_spmm`` (stc_csr_spmm_bf16 / stc_bcsr_spmm_bf16): X, Y0, Y bfloat16 with F % 8 == 0,
        graph values fp32, fp32 sums, one rounding at the end."""
        B, nc, F = X.shape
        self._bf16('spmm_bf16.X', X, (B, n_cols, F))
        self._bf16('spmm_bf16.Y', Y, (B, n_rows, F))
        if Y0 is not None:
            self._bf16('spmm_bf16.Y0', Y0, (B, n_rows, F))
        if F % 8:
            raise StcError(f'spmm_bf16: F = {F} must be a multiple of 8')
        self._i32('spmm.rowptr', rowptr, n_rows + 1)
        self._i32('spmm.colidx', colidx)
        self._f32('spmm.val', val, (colidx.numel(),))
        self._same_device(rowptr, colidx, val, X, Y0, Y)
        plain = Y0 is None or beta == 0
        nbytes = colidx.numel() * 8 + 4 * (n_rows + 1) + (2 if plain else 3) * 2 * B * n_rows * F
        tag = 'plain' if plain else 'with_y0'
        if (F % 512 == 0 and all(t is None or t.data_ptr() % 16 == 0 for t in (X, Y0, Y))
                and self._patch_wanted(plan, plan[3][3].shape[0] * B if plan is not None and len(plan) > 3 else 0, F > 1024, plain)):
            self._launch('stc_patch_spmm_bf16', X, *self._patch_ptrs(plan[3], X), n_rows, n_cols, _ptr(X), _ptr(Y0), _ptr(Y), B, F, float(alpha), float(beta),
                         nbytes=nbytes, tag=tag)
            return
        if plan is not None:
            blk_ptr, blk_cols, blk_vals = plan[:3]
            self._i32('spmm.blk_ptr', blk_ptr, (n_rows + 3) // 4 + 1)
            self._i32('spmm.blk_cols', blk_cols)
            self._f32('spmm.blk_vals', blk_vals, (blk_cols.numel(), 4))
            self._launch('stc_bcsr_spmm_bf16', X, _ptr(blk_ptr), _ptr(blk_cols), _ptr(blk_vals), n_rows, n_cols,
                         _ptr(X), _ptr(Y0), _ptr(Y), B, F, float(alpha), float(beta), nbytes=nbytes, tag=tag)
            return
        self._launch('stc_csr_spmm_bf16', X, _ptr(rowptr), _ptr(colidx), _ptr(val), n_rows, n_cols, _ptr(X), _ptr(Y0), _ptr(Y), B, F,
                     float(alpha), float(beta), nbytes=nbytes, tag=tag)

    def _patch_ptrs(self, patches, on):
        """Checked arguments (five arrays, n_patches, width) of the patch-form entry points (include/stc_hip.h, stc_patch_spmm_f32)."""
        pt_src, pt_rows, pt_cnt, pt_idx, pt_val = patches
        n_p, width = pt_idx.shape[0], pt_idx.shape[2]
        self._i32('spmm.pt_src', pt_src, n_p * PATCH_MAX_SRC)
        self._i32('spmm.pt_rows', pt_rows, n_p * PATCH_ROWS)
        self._i32('spmm.pt_cnt', pt_cnt, n_p * PATCH_ROWS)
        if pt_idx.dtype != torch.uint8 or not pt_idx.is_contiguous() or pt_idx.shape != (n_p, PATCH_ROWS, width):
            raise StcError(f'spmm.pt_idx must be a contiguous uint8 tensor of shape {(n_p, PATCH_ROWS, width)}')
        self._f32('spmm.pt_val', pt_val, (n_p, PATCH_ROWS, width))
        self._same_device(on, pt_src, pt_rows, pt_cnt, pt_idx, pt_val)
        return [_ptr(pt_src), _ptr(pt_rows), _ptr(pt_cnt), _ptr(pt_idx), _ptr(pt_val), n_p, width]

    def _patch_wanted(self, plan, items, wide, plain):
        """The launch-size rule of the patch form (see ``csr_spmm``): ``items`` = (patch, sample) workgroups, ``wide`` = rows of more than 2 KiB."""
        return plan is not None and len(plan) > 3 and self.patch_spmm and items >= self.patch_min_items * ((6 if wide else 3) if plain else 1)

    def _graph_ptrs(self, rowptr, colidx, val, plan, n_rows):
        self._i32('spmm.rowptr', rowptr, n_rows + 1)
        self._i32('spmm.colidx', colidx)
        self._f32('spmm.val', val, (colidx.numel(),))
        if plan is None:
            return [_ptr(rowptr), _ptr(colidx), _ptr(val), None, None, None]
        blk_ptr, blk_cols, blk_vals = plan[:3]
        self._i32('spmm.blk_ptr', blk_ptr, (n_rows + 3) // 4 + 1)
        self._i32('spmm.blk_cols', blk_cols)
        self._f32('spmm.blk_vals', blk_vals, (blk_cols.numel(), 4))
        return [_ptr(rowptr), _ptr(colidx), _ptr(val), _ptr(blk_ptr), _ptr(blk_cols), _ptr(blk_vals)]

    def _cell_rows(self, what, X, Y0, C, cin, h):
        """Shapes shared by the two fused backward products: X/Y0 (B, n, C*(cin+h+pad)); returns (B, n, pad)."""
        B, n, F = X.shape
        self._f32(what + '.X', X)
        self._f32(what + '.Y0', Y0, (B, n, F))
        if F % C or F // C < cin + h:
            raise StcError(f'{what}: row of {F} floats is not C={C} x (cin={cin} + h={h} + pad)')
        return B, n, F // C - cin - h

    def spmm_sum(self, rowptr, colidx, val, plan, X, X2, addends, Y, blend=None, alpha=1.0, amax=None):
        """Y = sum(scale * addend) + alpha * S.(X [+ X2]) on (B, n, C, h) state tensors (stc_spmm_sum_f32).  ``addends``: up to
        eight (tensor, column offset[, scale]) entries -- columns [off, off + h) of a (B, n, C, ld) tensor (a plain plane:
        ld = h, off = 0); scale defaults to 1.  ``amax``: float32 device tensor of slots, ZERO on entry; afterwards its maximum is
        max |Y| (an option kept for callers: the cell backward kernels find their gradient scales themselves since ABI v22)."""
        B, n, Cc, h = Y.shape
        self._f32('spmm_sum.Y', Y)
        self._f32('spmm_sum.X', X, (B, n, Cc, h))
        if X2 is not None:
            self._f32('spmm_sum.X2', X2, (B, n, Cc, h))
        if len(addends) > SPMM_SUM_MAX_ADD:
            raise StcError(f'spmm_sum: at most {SPMM_SUM_MAX_ADD} addends, got {len(addends)}')
        ptrs, lds, offs, scales = (_p * SPMM_SUM_MAX_ADD)(), (_i32 * SPMM_SUM_MAX_ADD)(), (_i32 * SPMM_SUM_MAX_ADD)(), (_f32 * SPMM_SUM_MAX_ADD)()
        for i, ent in enumerate(addends):
            t, off = ent[0], ent[1]
            self._f32(f'spmm_sum.add{i}', t)
            if t.shape[:3] != (B, n, Cc) or off < 0 or off + h > t.shape[-1] or (t.shape[-1] | off) & 3:
                raise StcError(f'spmm_sum: addend {i} of shape {tuple(t.shape)} / offset {off} does not fit')
            ptrs[i], lds[i], offs[i], scales[i] = t.data_ptr(), t.shape[-1], off, (float(ent[2]) if len(ent) > 2 else 1.0)
        U = Cand = dY = None
        if blend is not None:                                     # (U, Cand, dY): also dY = Y * U * (1 - Cand^2)
            U, Cand, dY = blend
            for name, t in (('U', U), ('Cand', Cand), ('dY', dY)):
                self._f32('spmm_sum.' + name, t, (B, n, Cc, h))
        if amax is not None:
            self._f32('spmm_sum.amax', amax)
        self._same_device(rowptr, colidx, val, X, X2, Y, U, Cand, dY, amax, *[ent[0] for ent in addends])
        g = self._graph_ptrs(rowptr, colidx, val, plan, n)
        self._launch('stc_spmm_sum_f32', Y, *g, n, n, _ptr(X), _ptr(X2), float(alpha), len(addends), ptrs, lds, offs, scales, _ptr(Y), _ptr(U), _ptr(Cand), _ptr(dY),
                     _ptr(amax), 0 if amax is None else amax.numel(), B, Cc, h,
                     nbytes=colidx.numel() * 8 + 4 * (n + 1) + 4 * B * n * Cc * h * (2 + (X2 is not None) + len(addends) + (3 if blend else 0)))

    RING2_MAX_ADD = 5

    @staticmethod
    def ring2_fits(B, n, Cc, h) -> bool:
        """Whether planes of this size are within the two-ring kernels' 32-bit piece offsets (< 2^28 sixteen-byte pieces) and chunking."""
        return h == 16 and (Cc * h) % 128 == 0 and B * n * (Cc * h // 4) < (1 << 28) and B <= 65535

    def ring2_sum(self, rowptr, colidx, val, ring2, X, X2, addends, U, Cand, Y, Z):
        """Y = sum(addends) + S.(X [+ X2]) and Z = S.(Y * U * (1 - Cand^2)) on (B, n, C, 16) state tensors in one launch (stc_ring2_sum_f32):
        the state-gradient sum with its blend backward AND the transpose aggregation of the candidate's gradient, without the dY plane.
        ``ring2`` = (l2_rows, l1_rows, int_rows, t1, t2) of ``graph._ring2_plan`` for S; the CSR arrays are what the CPU twin uses."""
        B, n, Cc, h = Y.shape
        for name, t in (('Y', Y), ('Z', Z), ('X', X), ('U', U), ('Cand', Cand)) + ((('X2', X2),) if X2 is not None else ()):
            self._f32('ring2_sum.' + name, t, (B, n, Cc, h))
        if len(addends) > self.RING2_MAX_ADD:
            raise StcError(f'ring2_sum: at most {self.RING2_MAX_ADD} addends, got {len(addends)}')
        ptrs = (_p * self.RING2_MAX_ADD)()
        for i, t in enumerate(addends):
            self._f32(f'ring2_sum.add{i}', t, (B, n, Cc, h))
            ptrs[i] = t.data_ptr()
        self._same_device(X, X2, U, Cand, Y, Z, *addends)
        self._launch('stc_ring2_sum_f32', Y, *self._ring2_ptrs('ring2_sum', ring2, Y), n, _ptr(X), _ptr(X2), len(addends), ptrs,
                     _ptr(U), _ptr(Cand), _ptr(Y), _ptr(Z), B, Cc, h,
                     nbytes=colidx.numel() * 8 + 4 * (n + 1) + 4 * B * n * Cc * h * (1 + (X2 is not None) + len(addends) + 2 + 2))

    def _ring2_ptrs(self, what, ring2, on):
        l2, l1, own, t1, t2 = ring2
        n_p = l2.shape[0]
        for name, t, shape in (('l2_rows', l2, (n_p, 96)), ('l1_rows', l1, (n_p, 64)), ('int_rows', own, (n_p, 32)), ('t1', t1, (n_p, 64, 8, 2)), ('t2', t2, (n_p, 32, 8, 2))):
            if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous() or tuple(t.shape) != shape:
                raise StcError(f'{what}.{name}: expected a contiguous int32 ROCm tensor of shape {shape}')
        self._same_device(on, l2, l1, own, t1, t2)
        return [_ptr(l2), _ptr(l1), _ptr(own), _ptr(t1), _ptr(t2), n_p]

    def ring2_blend(self, rowptr, colidx, val, ring2, Bm, A, U, H, Cand, Hnew, SHnew):
        """Cand = tanh(A + S.Bm), Hnew = (1 - U) H + U Cand and SHnew = S.Hnew in one launch (stc_ring2_blend_f32): ``spmm_blend_fwd`` without state
        copies + the plain aggregation of the new state, which is summed out of LDS instead of being read back.  ``ring2``: the plan for S."""
        B, n, Cc, h = H.shape
        for name, t in (('Bm', Bm), ('A', A), ('U', U), ('H', H), ('Cand', Cand), ('Hnew', Hnew), ('SHnew', SHnew)):
            self._f32('ring2_blend.' + name, t, (B, n, Cc, h))
        self._same_device(Bm, A, U, H, Cand, Hnew, SHnew)
        self._launch('stc_ring2_blend_f32', H, *self._ring2_ptrs('ring2_blend', ring2, H), n, _ptr(Bm), _ptr(A), _ptr(U), _ptr(H), _ptr(Cand), _ptr(Hnew), _ptr(SHnew),
                     B, Cc, h, nbytes=colidx.numel() * 8 + 4 * (n + 1) + 4 * B * n * Cc * h * 7)

    def ring2_chain(self, rowptr, colidx, val, ring2, X, X2, alpha1, add1, V, alpha2, add0, Z):
        """V = alpha1 S.(X [+ X2]) + sum(add1) and Z = alpha2 S.V + sum(scale * t for (t, scale) in add0) in one launch (stc_ring2_chain_f32): the
        order-3 feature recurrence [S.X, 2 S.(S.X) - X] and its transpose d0 - d2 + S^T (d1 + 2 S^T d2).  V may be None (not stored)."""
        B, n, Cc, h = Z.shape
        for name, t in (('Z', Z), ('X', X)) + ((('X2', X2),) if X2 is not None else ()) + ((('V', V),) if V is not None else ()):
            self._f32('ring2_chain.' + name, t, (B, n, Cc, h))
        if len(add1) > 2 or not 1 <= len(add0) <= self.RING2_MAX_ADD:
            raise StcError(f'ring2_chain: 0..2 first-ring and 1..{self.RING2_MAX_ADD} interior addends, got {len(add1)} and {len(add0)}')
        p1, p0, s0 = (_p * 2)(), (_p * self.RING2_MAX_ADD)(), (_f32 * self.RING2_MAX_ADD)()
        for i, t in enumerate(add1):
            self._f32(f'ring2_chain.add1[{i}]', t, (B, n, Cc, h))
            p1[i] = t.data_ptr()
        for i, (t, scale) in enumerate(add0):
            self._f32(f'ring2_chain.add0[{i}]', t, (B, n, Cc, h))
            p0[i], s0[i] = t.data_ptr(), float(scale)
        self._same_device(X, X2, V, Z, *add1, *[t for t, _ in add0])
        self._launch('stc_ring2_chain_f32', Z, *self._ring2_ptrs('ring2_chain', ring2, Z), n, _ptr(X), _ptr(X2), float(alpha1), len(add1), p1, _ptr(V),
                     float(alpha2), len(add0), p0, s0, _ptr(Z), B, Cc, h,
                     nbytes=colidx.numel() * 8 + 4 * (n + 1) + 4 * B * n * Cc * h * (1 + (X2 is not None) + len(add1) + (V is not None) + len(add0) + 1))

    def csr_sddmm(self, rowptr, colidx, n_rows, n_cols, A, Bm, out, alpha, accumulate):
        B, nr, F = A.shape
        self._f32('sddmm.A', A, (B, n_rows, F))
        self._f32('sddmm.Bm', Bm, (B, n_cols, F))
        self._i32('sddmm.rowptr', rowptr, n_rows + 1)
        self._i32('sddmm.colidx', colidx)
        self._f32('sddmm.out', out, (colidx.numel(),))
        self._same_device(rowptr, colidx, A, Bm, out)
        self._launch('stc_csr_sddmm_f32', A, _ptr(rowptr), _ptr(colidx), n_rows, n_cols, _ptr(A), _ptr(Bm), _ptr(out), B, F, float(alpha), int(bool(accumulate)))

    # ---- category graph --------------------------------------------------------------
    def cheby_dense_fwd(self, G, K, T):
        n = G.shape[0]
        self._f32('cheby.G', G, (n, n))
        self._f32('cheby.T', T, (K, n, n))
        self._launch('stc_cheby_dense_fwd_f32', G, _ptr(G), n, K, _ptr(T))

    def cheby_dense_bwd(self, G, T, dT, dG):
        K, n, _ = T.shape
        self._f32('cheby.G', G, (n, n))
        self._f32('cheby.T', T, (K, n, n))
        self._f32('cheby.dT', dT, (K, n, n))
        self._f32('cheby.dG', dG, (n, n))
        self._launch('stc_cheby_dense_bwd_f32', G, _ptr(G), _ptr(T), _ptr(dT), n, K, _ptr(dG))

    # ---- node kernel -------------------------------------------------------------------
    @staticmethod
    def _ptr_array(tensors):
        arr = (_p * len(tensors))()
        for i, t in enumerate(tensors):
            arr[i] = t.data_ptr()
        return arr

    def _node_shapes(self, Zs, Tc, W):
        Ks, Kc = len(Zs), Tc.shape[0]
        if not (1 <= Ks <= MAX_K and 1 <= Kc <= MAX_K):
            raise StcError(f'Chebyshev orders Ks={Ks}, Kc={Kc} outside [1,{MAX_K}]')
        R, Cc, L = Zs[0].shape
        Ho = W.shape[1]
        Lw = W.shape[0] // (Ks * Kc)            # feature rows per W block; slab columns [Lw, L) are zero padding
        if Lw < 1 or Lw > L:
            raise StcError(f'node.W: {W.shape[0]} rows give Lw={Lw} per block, slabs are {L} wide')
        for i, z in enumerate(Zs):
            self._f32(f'node.Z[{i}]', z, (R, Cc, L))
        self._f32('node.Tc', Tc, (Kc, Cc, Cc))
        self._f32('node.W', W, (Ks * Kc * Lw, Ho))
        return Ks, Kc, R, Cc, L, Lw, Ho

    def bdg_node_fwd(self, Zs: Sequence[torch.Tensor], Tc, W, bias, Y):
        if Zs[0].dtype == torch.bfloat16:
            return self.bdg_node_fwd_bf16(Zs, Tc, W, bias, Y)
        Ks, Kc, R, Cc, L, Lw, Ho = self._node_shapes(Zs, Tc, W)
        if bias is not None:
            self._f32('node.bias', bias, (Ho,))
        self._f32('node.Y', Y, (R, Cc, Ho))
        self._same_device(*Zs, Tc, W, bias, Y)
        self._launch('stc_bdg_node_fwd_f32', Y, self._ptr_array(Zs), Ks, _ptr(Tc), Kc, _ptr(W), _ptr(bias), _ptr(Y), R, Cc, L, Lw, Ho,
                     nbytes=4 * R * Cc * (Ks * L + Ho))          # the Ks slabs in, Y out

    def _get_workspace(self, device, nbytes):
        """Scratch for the backward kernels' per-workgroup partial sums: one buffer per (device, stream) -- launches on two
        streams never share one -- sized up front for the largest shape on the matrix-core paths, so that it is not replaced
        in practice; if a larger request does come, the old buffer is KEPT alive (a captured HIP graph or a launch still in
        flight may hold its address) and a new one is used from then on."""
        key = (device, torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch._C._cuda_getDevice()))
        ws = self._workspace.get(key)
        if ws is None or ws.numel() < nbytes:
            if ws is not None:
                self._retired.append(ws)
            floor = self.lib.stc_bdg_node_bwd_workspace_bytes(3, 3, 64, 32, 32, 1)      # K = 3, C = 64, L = Ho = 32, dTc wanted
            # persistent scratch: survives the trainer's torch.cuda.empty_cache() after every step
            ws = torch.empty(max(nbytes, floor, 1 << 20), dtype=torch.uint8, device=device)
            self._workspace[key] = ws
        return ws

    def bdg_node_bwd(self, Zs, Tc, W, dY, dZs, dW, db, dTc):
        if Zs[0].dtype == torch.bfloat16:
            if dTc is not None:
                raise StcError('bdg_node_bwd: bf16 slabs are for fixed category graphs (no dTc)')
            return self.bdg_node_bwd_bf16(Zs, Tc, W, dY, dZs, dW, db)
        Ks, Kc, R, Cc, L, Lw, Ho = self._node_shapes(Zs, Tc, W)
        self._f32('node.dY', dY, (R, Cc, Ho))
        if len(dZs) != Ks:
            raise StcError('node.dZ: need one gradient slab per Chebyshev order')
        for i, z in enumerate(dZs):
            self._f32(f'node.dZ[{i}]', z, (R, Cc, L))
        self._f32('node.dW', dW, (Ks * Kc * Lw, Ho))
        if db is not None:
            self._f32('node.db', db, (Ho,))
        if dTc is not None:
            self._f32('node.dTc', dTc, (Kc, Cc, Cc))
        self._same_device(*Zs, Tc, W, dY, *dZs, dW, db, dTc)
        nbytes = self.lib.stc_bdg_node_bwd_workspace_bytes(Ks, Kc, Cc, L, Ho, int(dTc is not None))
        ws = self._get_workspace(dY.device, nbytes)
        self._launch('stc_bdg_node_bwd_f32', dY, self._ptr_array(Zs), Ks, _ptr(Tc), Kc, _ptr(W), _ptr(dY), self._ptr_array(dZs), _ptr(dW), _ptr(db), _ptr(dTc), _ptr(ws), ws.numel(), R, Cc, L, Lw, Ho,
                     nbytes=4 * R * Cc * (2 * Ks * L + Ho))      # the Ks slabs and dY in, the Ks gradient slabs out

    def mix_dT_supported(self, Ks, Kc, Cc, L, Ho) -> bool:
        return bool(self.lib.stc_mix_dt_supported(Ks, Kc, Cc, L, Ho))

    def mix_dT(self, Zs, W, dY, dTc):
        """dTc (Kc, C, C) = the category graph's gradient through one BDG_Dif for few categories (stc_mix_dt_f32): Zs = the Ks slabs (R, C, L) with
        R * C a multiple of 16, dY (R, C, Ho), W (Ks * Kc * Lw, Ho)."""
        Ks = len(Zs)
        R, Cc, L = Zs[0].shape
        Ho = W.shape[1]
        Lw = W.shape[0] // (Ks * Ks)
        for i, z in enumerate(Zs):
            self._f32(f'mix_dT.Z[{i}]', z, (R, Cc, L))
        self._f32('mix_dT.W', W, (Ks * Ks * Lw, Ho))
        self._f32('mix_dT.dY', dY, (R, Cc, Ho))
        self._f32('mix_dT.dTc', dTc, (Ks, Cc, Cc))
        self._same_device(*Zs, W, dY, dTc)
        ws = self._get_workspace(dY.device, self.lib.stc_mix_dt_workspace_bytes(Ks))
        self._launch('stc_mix_dt_f32', dY, self._ptr_array(Zs), Ks, _ptr(W), _ptr(dY), _ptr(dTc), _ptr(ws), ws.numel(), R * Cc, Cc, L, Lw, Ho,
                     nbytes=4 * R * Cc * (Ks * L + Ho))

    # ---- bf16 storage (configuration 5) ------------------------------------------------------------
    def node_bf16_supported(self, Ks, Kc, Cc, L, Ho) -> bool:
        return bool(self.lib.stc_bdg_node_bf16_supported(Ks, Kc, Cc, L, Ho))

    def _node_shapes_bf16(self, Zs, Tc, W):
        Ks, Kc = len(Zs), Tc.shape[0]
        R, Cc, L = Zs[0].shape
        Ho = W.shape[1]
        Lw = W.shape[0] // (Ks * Kc)
        if Lw < 1 or Lw > L or W.shape[0] != Ks * Kc * Lw:
            raise StcError(f'node_bf16.W: {W.shape[0]} rows give Lw={Lw} per block, slabs are {L} wide')
        for i, z in enumerate(Zs):
            self._bf16(f'node_bf16.Z[{i}]', z, (R, Cc, L))
        self._f32('node_bf16.Tc', Tc, (Kc, Cc, Cc))
        self._f32('node_bf16.W', W, (Ks * Kc * Lw, Ho))
        return Ks, Kc, R, Cc, L, Lw, Ho

    def bdg_node_fwd_bf16(self, Zs: Sequence[torch.Tensor], Tc, W, bias, Y):
        """bf16 slabs / output, fp32 weights (stc_bdg_node_fwd_bf16)."""
        Ks, Kc, R, Cc, L, Lw, Ho = self._node_shapes_bf16(Zs, Tc, W)
        if bias is not None:
            self._f32('node_bf16.bias', bias, (Ho,))
        self._bf16('node_bf16.Y', Y, (R, Cc, Ho))
        self._same_device(*Zs, Tc, W, bias, Y)
        self._launch('stc_bdg_node_fwd_bf16', Y, self._ptr_array(Zs), Ks, _ptr(Tc), Kc, _ptr(W), _ptr(bias), _ptr(Y), R, Cc, L, Lw, Ho,
                     nbytes=2 * R * Cc * (Ks * L + Ho))

    def bdg_node_bwd_bf16(self, Zs, Tc, W, dY, dZs, dW, db):
        Ks, Kc, R, Cc, L, Lw, Ho = self._node_shapes_bf16(Zs, Tc, W)
        self._bf16('node_bf16.dY', dY, (R, Cc, Ho))
        if len(dZs) != Ks:
            raise StcError('node_bf16.dZ: need one gradient slab per Chebyshev order')
        for i, z in enumerate(dZs):
            self._bf16(f'node_bf16.dZ[{i}]', z, (R, Cc, L))
        self._f32('node_bf16.dW', dW, (Ks * Kc * Lw, Ho))
        if db is not None:
            self._f32('node_bf16.db', db, (Ho,))
        self._same_device(*Zs, Tc, W, dY, *dZs, dW, db)
        nbytes = self.lib.stc_bdg_node_bwd_workspace_bytes(Ks, Kc, Cc, L, Ho, 0)
        ws = self._get_workspace(dY.device, nbytes)
        self._launch('stc_bdg_node_bwd_bf16', dY, self._ptr_array(Zs), Ks, _ptr(Tc), Kc, _ptr(W), _ptr(dY), self._ptr_array(dZs), _ptr(dW), _ptr(db),
                     _ptr(ws), ws.numel(), R, Cc, L, Lw, Ho, nbytes=2 * R * Cc * (2 * Ks * L + Ho))

    @property
    def bf16(self):
        """The planar-cell kernel set for bf16 planes (``_Bf16Planar``): what ``ops.stc_cell_graph`` launches through when its
        state tensors are bfloat16."""
        if getattr(self, '_bf16_front', None) is None:
            self._bf16_front = _Bf16Planar(self)
        return self._bf16_front

    # ---- post-aggregation form (Ks = Kc = 2): Y = A + S.Bm --------------------------------------
    def node_post_supported(self, Ks, Kc, Cc, L, Ho) -> bool:
        return bool(self.lib.stc_bdg_node_post_supported(Ks, Kc, Cc, L, Ho))

    def _post_rows(self, X, X2, Tc, W):
        """Shapes of the post-aggregation kernels; X2 given: planar rows, X and X2 are the two (R, C, 16) planes."""
        if X2 is None:
            return self._node_shapes([X, X], Tc, W)
        self._f32('post.X', X)
        if X.dim() != 3 or X.shape[-1] != 16:
            raise StcError(f'post: the leading planar input plane must be (rows, C, 16), got {tuple(X.shape)}')
        R, Cc, _ = X.shape
        w2 = X2.shape[-1]
        self._f32('post.X2', X2, (R, Cc, w2))
        if not (w2 == 16 or 1 <= w2 <= 4):
            raise StcError(f'post: second plane must be 16 or 1..4 columns wide, got {w2}')
        if Tc.dim() != 3 or Tc.shape[0] != 2 or W.dim() != 2 or W.shape[0] != 4 * (16 + w2):
            raise StcError(f'post: planar form needs Ks = Kc = 2 and W with 4 x {16 + w2} rows, got {tuple(W.shape)}')
        self._f32('post.Tc', Tc, (2, Cc, Cc))
        self._f32('post.W', W)
        return 2, 2, R, Cc, (32 if w2 == 16 else 20), 16 + w2, W.shape[1]

    def node_post_fwd(self, X, Tc, W, bias, A, Bm, X2=None):
        """X -> A = sum_c T_c^T (X W_{0,c}) + bias, Bm = sum_c T_c^T (X W_{1,c}); the caller finishes Y = A + S.Bm."""
        Ks, Kc, R, Cc, L, Lw, Ho = self._post_rows(X, X2, Tc, W)
        if bias is not None:
            self._f32('post.bias', bias, (Ho,))
        for name, t in (('A', A), ('Bm', Bm)):
            self._f32('post.' + name, t, (R, Cc, Ho))
        self._same_device(X, X2, Tc, W, bias, A, Bm)
        self._launch('stc_bdg_node_post_fwd_f32', X, _ptr(X), _ptr(X2), _ptr(Tc), _ptr(W), _ptr(bias), _ptr(A), _ptr(Bm), R, Cc, L, Lw, Ho)

    def spmm_blend_fwd(self, rowptr, colidx, val, plan, Bm, A, U, H, Cand, Hnew, copies=(), side=None):
        """Y = A + S.Bm with the GRU blend in the epilogue (stc_spmm_blend_fwd_f32).  Bm/A/U/H/Cand/Hnew (B, n, C, h);
        ``copies`` / ``side`` as in ``cell_blend_fwd`` (buffers (B*n, C, ld))."""
        B, n, Cc, h = H.shape
        for name, t in (('Bm', Bm), ('A', A), ('U', U), ('H', H), ('Cand', Cand), ('Hnew', Hnew)):
            self._f32('spmm_blend.' + name, t, (B, n, Cc, h))
        if len(copies) > 2 or (side is not None and not copies):
            raise StcError('spmm_blend: at most two state copies; side needs a first copy')
        cp = []
        for i, (buf, off) in enumerate(copies):
            self._f32(f'spmm_blend.copy{i}', buf)
            if buf.dim() != 3 or buf.shape[:2] != (B * n, Cc) or off < 0 or off + h > buf.shape[-1]:
                raise StcError(f'spmm_blend: copy{i} of shape {tuple(buf.shape)} cannot take columns [{off}, {off + h})')
            cp.append((buf, buf.shape[-1], off))
        while len(cp) < 2:
            cp.append((None, 0, 0))
        side_cin = 0
        if side is not None:
            side_cin = side.shape[-1]
            self._f32('spmm_blend.side', side, (B * n, Cc, side_cin))
            if side_cin != cp[0][2]:
                raise StcError(f'spmm_blend: side width {side_cin} must equal the first copy\'s column offset {cp[0][2]}')
        self._same_device(rowptr, colidx, val, Bm, A, U, H, Cand, Hnew, cp[0][0], cp[1][0], side)
        g = self._graph_ptrs(rowptr, colidx, val, plan, n)
        self._launch('stc_spmm_blend_fwd_f32', H, *g, n, n, _ptr(Bm), _ptr(A), _ptr(U), _ptr(H), _ptr(Cand), _ptr(Hnew),
                     _ptr(cp[0][0]), cp[0][1], cp[0][2], _ptr(side), side_cin, _ptr(cp[1][0]), cp[1][1], cp[1][2], B, Cc, h,
                     nbytes=colidx.numel() * 8 + 4 * (n + 1) + 4 * B * n * Cc * h * (6 + len(copies)))

    def node_post_bwd(self, X, Tc, W, dA, dB, dX, dW, db, X2=None, dX2=None, act_amax=None):
        """(X, dA = dY, dBm = S^T dY) -> dX, dW, db of the convolution in its post-aggregation form.  Planar (X2 given):
        the gradient comes out as the two planes dX, dX2 as well.  fp16 x 2 format (planar, C = 64): ``act_amax`` = (slots of max |X|, slots of
        max |X2|), rows of what a forward launch left (R*H takes H's)."""
        Ks, Kc, R, Cc, L, Lw, Ho = self._post_rows(X, X2, Tc, W)
        f16 = X2 is not None and Cc == 64
        narrow = X2 is not None and X2.shape[-1] != 16
        if (dX2 is not None) != (X2 is not None and not narrow):
            raise StcError('post: a planar gradient (dX2) goes with a 16 + 16 planar input and only with it')
        for name, t in (('dA', dA), ('dB', dB)):
            self._f32('post.' + name, t, (R, Cc, Ho))
        self._f32('post.dX', dX, (R, Cc, L) if X2 is None else (R, Cc, 16))
        if dX2 is not None:
            self._f32('post.dX2', dX2, (R, Cc, 16))
        self._f32('post.dW', dW, (Ks * Kc * Lw, Ho))
        if db is not None:
            self._f32('post.db', db, (Ho,))
        self._same_device(X, X2, Tc, W, dA, dB, dX, dX2, dW, db)
        ws = self._get_workspace(X.device, self.lib.stc_bdg_node_bwd_workspace_bytes(Ks, Kc, Cc, L, Ho, 0))
        self._launch('stc_bdg_node_post_bwd_f32', X, _ptr(X), _ptr(X2), _ptr(Tc), _ptr(W), _ptr(dA), _ptr(dB), _ptr(dX), _ptr(dX2), _ptr(dW), _ptr(db),
                     self.operand_format if f16 else FMT_BF16X3,      # (the forms built for fp16 x 2)
                     self._act_amax('post.x', None if act_amax is None else act_amax[0].view(1, -1), 1, X) if f16 else None,
                     self._act_amax('post.x2', None if act_amax is None else act_amax[1].view(1, -1), 1, X) if f16 else None,
                     _ptr(ws), ws.numel(), R, Cc, L, Lw, Ho,
                     nbytes=4 * R * Cc * ((16 if X2 is not None else L) + (0 if X2 is None else X2.shape[-1]) + 2 * Ho + (16 if X2 is not None else L) + (16 if dX2 is not None else 0)))

    # ---- planar cell inputs (Ks = Kc = 2, cin = h = 16) ---------------------------------------------
    def cell_planar_supported(self, Ks, Kc, Cc, h) -> bool:
        return bool(self.lib.stc_cell_planar_supported(Ks, Kc, Cc, h))

    def _planes(self, what, X, H, SX, SH):
        """State planes (R, C, h); input planes (R, C, cin) with cin = h or 1..4 (narrow: layer 0)."""
        R, Cc, h = H.shape
        cin = X.shape[-1]
        if not (cin == h or 1 <= cin <= 4):
            raise StcError(f'{what}: input plane width {cin} must be {h} or 1..4')
        for name, t in (('H', H), ('SH', SH)):
            self._f32(f'{what}.{name}', t, (R, Cc, h))
        for name, t in (('X', X), ('SX', SX)):
            self._f32(f'{what}.{name}', t, (R, Cc, cin))
        return R, Cc, h, cin

    def cell_planar_post_fused(self, Cc) -> bool:
        """Whether cell_gates_fwd_planar can also run the candidate's projection (``post=``) for this category count."""
        return Cc in (32, 64)

    def cell_gates_fwd_planar(self, X, H, SX, SH, Tc, W, bias, U, Rg, RH, post=None, act_amax=None):
        """Gates convolution on planar inputs; writes U, Rg and the R*H plane (the candidate's input is (X, RH)).
        ``post`` = (Wc, bc, A, Bm): the same launch also writes the candidate's post-aggregation pair A, Bm; ``RH`` may then be
        None (not written: ``cell_bwd_planar`` forms R*H itself).  ``act_amax`` (fp16 x 2 format): (4, 256) zero floats that receive the
        maxima of the four input planes -- what the backward launches scale their activation operands by (``act_amax_buffer``)."""
        R, Cc, h, cin = self._planes('planar', X, H, SX, SH)
        if RH is None and post is None:
            raise StcError('planar gates: the R*H plane is optional only with the fused candidate projection (post=)')
        self._f32('planar.Tc', Tc, (2, Cc, Cc))
        self._f32('planar.W', W)
        if W.shape != (4 * (cin + h), 2 * h):
            raise StcError(f'planar gates: W {tuple(W.shape)} is not ({4 * (cin + h)}, {2 * h})')
        if bias is not None:
            self._f32('planar.bias', bias, (2 * h,))
        for name, t in (('U', U), ('Rg', Rg)) + ((('RH', RH),) if RH is not None else ()):
            self._f32('planar.' + name, t, (R, Cc, h))
        Wc = bc = A = Bm = None
        if post is not None:
            Wc, bc, A, Bm = post
            self._f32('planar.Wc', Wc, (4 * (cin + h), h))
            if bc is not None:
                self._f32('planar.bc', bc, (h,))
            for name, t in (('A', A), ('Bm', Bm)):
                self._f32('planar.' + name, t, (R, Cc, h))
        self._same_device(X, H, SX, SH, Tc, W, bias, U, Rg, RH, Wc, bc, A, Bm)
        self._launch('stc_cell_gates_fwd_planar_f32', H, _ptr(X), _ptr(H), _ptr(SX), _ptr(SH), _ptr(Tc), _ptr(W), _ptr(bias),
                     _ptr(U), _ptr(Rg), _ptr(RH), _ptr(Wc), _ptr(bc), _ptr(A), _ptr(Bm), self.operand_format, self._act_amax('planar', act_amax, 4, H),
                     R, Cc, cin + h, h,
                     # algorithmic bytes: X, SX (cin wide), H, SH in; U, Rg (+ RH, + A, Bm) out -- every plane once
                     nbytes=4 * R * Cc * (2 * cin + 2 * h + h * (2 + (RH is not None) + (2 if post is not None else 0))))

    def cell_gates_bwd_planar(self, X, H, SX, SH, Tc, W, dRH, Cand, U, Rg, dHnew, dZs, dW, db, dH, act_amax=None):
        """``dRH``: gradient of the R*H plane; ``dZs`` = [d X plane, d SX plane, d H plane, d SH plane] (the first two
        None for a narrow input plane, which needs no gradient)."""
        R, Cc, h, cin = self._planes('planar', X, H, SX, SH)
        self._f32('planar.Tc', Tc, (2, Cc, Cc))
        self._f32('planar.W', W, (4 * (cin + h), 2 * h))
        for name, t in (('dRH', dRH), ('Cand', Cand), ('U', U), ('Rg', Rg), ('dHnew', dHnew)) + ((('dH', dH),) if dH is not None else ()):
            self._f32('planar.' + name, t, (R, Cc, h))      # dH None: the kernel folds the state's share into dZs[2]
        if len(dZs) != 4:
            raise StcError('planar gates backward: four gradient planes (dX, dSX, dH, dSH)')
        for i, z in enumerate(dZs):
            if z is None and i < 2 and cin != h:
                continue
            self._f32(f'planar.dZ[{i}]', z, (R, Cc, h))
        self._f32('planar.dW', dW, (4 * (cin + h), 2 * h))
        if db is not None:
            self._f32('planar.db', db, (2 * h,))
        self._same_device(X, H, SX, SH, Tc, W, dRH, Cand, U, Rg, dHnew, *dZs, dW, db, dH)
        ws = self._get_workspace(H.device, self.lib.stc_bdg_node_bwd_workspace_bytes(2, 2, Cc, 2 * h, 2 * h, 0))
        zp = (_p * 4)(*[0 if z is None else z.data_ptr() for z in dZs])
        self._launch('stc_cell_gates_bwd_planar_f32', H, _ptr(X), _ptr(H), _ptr(SX), _ptr(SH), _ptr(Tc), _ptr(W), _ptr(dRH), _ptr(Cand),
                     _ptr(U), _ptr(Rg), _ptr(dHnew), zp, _ptr(dW), _ptr(db), _ptr(dH), self.operand_format,
                     self._act_amax('planar', act_amax, 4, H), _ptr(ws), ws.numel(), R, Cc, cin + h, h,
                     nbytes=4 * R * Cc * (2 * cin + 2 * h + 5 * h + 2 * h + (2 * h if cin == h else 0) + (h if dH is not None else 0)))

    # ---- the whole backward of a planar cell step in one launch ----------------------------------------
    def cell_bwd_planar_supported(self, Cc, h) -> bool:
        return bool(self.lib.stc_cell_bwd_planar_supported(Cc, h))

    def cell_bwd_planar(self, X, H, SX, SH, Tc, Wg, Wc, U, Rg, Cand, dHnew, dBm, dZs, dWg, dbg, dWc, dbc, accumulate_x=False, accumulate_h=False,
                        act_amax=None):
        """Candidate (post-aggregation form) + gates backward of one planar cell step in one launch.  ``dBm`` = S^T dY with
        dY = dHnew U (1 - Cand^2) (the kernel re-forms dY itself); ``dZs`` = [dX, dSX, dH, dSH] gradient planes: dX = the candidate's
        plus the gates' share of the X plane, dH includes the gate prologue's share (dX, dSX None for a narrow input plane).
        ``accumulate_x`` / ``accumulate_h``: the X-side / H-side planes already hold the state's other consumer's gradients; add to them.
        ``act_amax`` (fp16 x 2 operand format): the (4, 256) plane maxima the forward launch (``cell_gates_fwd_planar(act_amax=)``) left for the
        same X, H, SX, SH; gradient scales the kernel finds itself."""
        R, Cc, h, cin = self._planes('cell_bwd', X, H, SX, SH)
        self._f32('cell_bwd.Tc', Tc, (2, Cc, Cc))
        self._f32('cell_bwd.Wg', Wg, (4 * (cin + h), 2 * h))
        self._f32('cell_bwd.Wc', Wc, (4 * (cin + h), h))
        for name, t in (('U', U), ('Rg', Rg), ('Cand', Cand), ('dHnew', dHnew), ('dBm', dBm)):
            self._f32('cell_bwd.' + name, t, (R, Cc, h))
        if len(dZs) != 4:
            raise StcError('cell backward: four gradient planes (dX, dSX, dH, dSH)')
        for i, z in enumerate(dZs):
            if z is None and i < 2 and cin != h:
                continue
            self._f32(f'cell_bwd.dZ[{i}]', z, (R, Cc, h))
        self._f32('cell_bwd.dWg', dWg, (4 * (cin + h), 2 * h))
        self._f32('cell_bwd.dWc', dWc, (4 * (cin + h), h))
        if dbg is not None:
            self._f32('cell_bwd.dbg', dbg, (2 * h,))
        if dbc is not None:
            self._f32('cell_bwd.dbc', dbc, (h,))
        self._same_device(X, H, SX, SH, Tc, Wg, Wc, U, Rg, Cand, dHnew, dBm, *dZs, dWg, dbg, dWc, dbc)
        ws = self._get_workspace(H.device, self.lib.stc_cell_bwd_planar_workspace_bytes(Cc, 2 * h, h))
        if accumulate_x and cin != h:
            raise StcError('cell backward: accumulate_x with a narrow input plane (it gets no gradient)')
        self._launch('stc_cell_bwd_planar_f32', H, _ptr(X), _ptr(H), _ptr(SX), _ptr(SH), _ptr(Tc), _ptr(Wg), _ptr(Wc), _ptr(U), _ptr(Rg), _ptr(Cand),
                     _ptr(dHnew), _ptr(dBm), *[_ptr(z) for z in dZs], _ptr(dWg), _ptr(dbg), _ptr(dWc), _ptr(dbc), int(bool(accumulate_x)), int(bool(accumulate_h)),
                     self.operand_format, self._act_amax('cell_bwd', act_amax, 4, H), _ptr(ws), ws.numel(), R, Cc, cin + h, h,
                     # algorithmic bytes: X, SX (cin wide), H, SH, U, Rg, Cand, dHnew, dBm in; dH, dSH (+ dX, dSX) out; a plane that is
                     # accumulated into is also read -- every plane once
                     nbytes=4 * R * Cc * (2 * cin + 7 * h + 2 * h * (1 + bool(accumulate_h)) + (2 * h * (1 + bool(accumulate_x)) if cin == h else 0)),
                     tag='wide' if cin == h else 'layer0')

    # ---- small graphs: one STC_Cell step per launch --------------------------------------------------------
    SMALL_MAX_ROWS = 65535       # N*C rows per sample the kernels take at all (16-bit row arithmetic)
    #: N*C rows per sample up to which the executor prefers these kernels over the general path (C <= 16 has no split-operand matrix-core
    #: kernels there).  Measured as HIP-graph replays at batch 32, these kernels | general path, ms per step: 500 rows (SF) 2.2 | 4.9;
    #: 1 568 (N = 196, C = 8) 2.9 | 7.6;  3 200: 4.5 | 11.5;  6 912 (C = 12): 8.8 | 22.4;  32 768 (C = 8): 37 | 90;  but 16 384 rows at
    #: C = 16: 18.3 | 13.4 (the general path's fp32-MFMA node kernels take C = 16) -- hence the extra rule in small.small_graph_supported.
    SMALL_PREFERRED_ROWS = 65535
    SMALL_STAGED_ROWS = 640      # N*C rows per sample that a compute unit's LDS stages (above: every gather from L2; dense graphs go to the general path)

    def cell_small_supported(self, Ks, Kc, Cc, cin, h, n_nodes=0) -> bool:
        return n_nodes * Cc <= self.SMALL_MAX_ROWS and bool(self.lib.stc_cell_small_supported(Ks, Kc, Cc, cin, h))

    @staticmethod
    def cell_small_zg_width(cin) -> int:
        """Floats per row of the saved aggregate ``Zg = S.[H | Xt | 0]``."""
        return 32 if cin == 16 else 20

    @staticmethod
    def cell_small_params(Ks, Kc, cin, h=16) -> int:
        """Floats per row of the parameter-gradient partials: [dWg | dbg | dWc | dbc]."""
        return Ks * Kc * (cin + h) * 3 * h + 3 * h

    @property
    def cell_small_param_rows(self) -> int:
        """Rows of the parameter-gradient partials per sample (one per wave of the backward's workgroup)."""
        return self.lib.stc_cell_small_param_rows()

    def _small_shapes(self, what, rowptr, colidx, val, X, H, Tc, Wg, Wc, planes, Zg, Zc):
        if H.dim() != 4 or X.dim() != 4 or X.shape[:3] != H.shape[:3]:
            raise StcError(f'{what}: X {tuple(X.shape)} / H {tuple(H.shape)} must be (B, N, C, cin) / (B, N, C, 16)')
        B, N, Cc, h = H.shape
        cin = X.shape[-1]
        Kc = Ks = Tc.shape[0]                                     # (the reference's layers take one order for both graphs: Main.py:24)
        if not self.cell_small_supported(Ks, Kc, Cc, cin, h, N):
            raise StcError(f'{what}: shape outside the small-graph cell kernels (Ks = Kc = 2 or 3, hidden 16, C <= 16, cin = 16 or 1..4, '
                           f'N*C <= {self.SMALL_MAX_ROWS}): Kc={Kc} C={Cc} cin={cin} h={h} N={N}')
        self._f32(what + '.X', X)
        self._f32(what + '.H', H)
        self._f32(what + '.Tc', Tc, (Kc, Cc, Cc))
        L = cin + h
        self._f32(what + '.Wg', Wg, (Ks * Kc * L, 2 * h))
        self._f32(what + '.Wc', Wc, (Ks * Kc * L, h))
        self._i32(what + '.rowptr', rowptr, N + 1)
        self._i32(what + '.colidx', colidx, val.numel())
        self._f32(what + '.val', val)
        for name, t in planes.items():
            self._f32(f'{what}.{name}', t, (B, N, Cc, h))
        self._f32(what + '.Zg', Zg, (B, N * Cc, self.cell_small_zg_width(cin)))
        self._f32(what + '.Zc', Zc, (B, N * Cc, h))
        self._same_device(H, X, Tc, Wg, Wc, rowptr, colidx, val, Zg, Zc, *planes.values())
        return B, N, Cc, cin, Kc

    def _small_order3(self, what, Tc, graph2, planes2, like_g, like_c, dense, N):
        """Arguments of the order-3 form: the second graph's CSR (T_2(S) in the launch's orientation) and the third-slab planes; order 2: nulls."""
        if Tc.shape[0] != 3:
            return (None, None, None, 0), (None, None)
        if graph2 is None or planes2 is None or any(t is None for t in planes2) or dense:
            raise StcError(f'{what}: Chebyshev order 3 takes graph2 = the CSR of T_2(S) = 2 S^2 - I (CsrGraph.second_order), the planes Zg2 / Zc2, and a fixed CSR graph')
        rp2, ci2, v2 = graph2
        self._i32(what + '.rowptr2', rp2, N + 1)
        self._i32(what + '.colidx2', ci2, v2.numel())
        self._f32(what + '.val2', v2)
        self._f32(what + '.Zg2', planes2[0], tuple(like_g.shape))
        self._f32(what + '.Zc2', planes2[1], tuple(like_c.shape))
        self._same_device(like_g, rp2, ci2, v2, *planes2)
        return (rp2.data_ptr(), ci2.data_ptr(), v2.data_ptr(), v2.numel()), (planes2[0].data_ptr(), planes2[1].data_ptr())

    def cell_small_fwd(self, rowptr, colidx, val, X, H, Tc, Wg, bg, Wc, bc, U, R, Cand, Hnew, RH, Zg, Zc, checked=True, Z0=None, splits=1, Z0c=None,
                       Z1c=None, graph2=None, Zg2=None, Zc2=None):
        """One STC_Cell step (reference STC_GNN.py:65-79) in one launch: ``stc_cell_small_fwd_f32``.  (rowptr, colidx, val): CSR of Gs^T.
        Chebyshev order 3 (``Tc`` of three matrices): ``graph2`` = (rowptr, colidx, val) of 2 (Gs^T)^2 - I and the planes ``Zg2`` (like Zg), ``Zc2`` (like Zc).
        ``checked=False``: the caller built every buffer itself from shapes it already validated (the cell-graph executor).
        ``Z0`` (optional, like Zg): receives the slab [H | Xt | 0] (learned graphs: operand of the graph-gradient product).
        ``splits`` = G > 1: the cell as TWO launches (phases 1 + 2, then 3 + 4: R*H of the neighbours is the one dependency that crosses
        workgroups), each over G workgroups per sample that own a contiguous range of row tiles -- for batches too small to fill the chip
        with one workgroup per sample (``cell_small_splits``)."""
        if checked:
            B, N, Cc, cin, Kc = self._small_shapes('cell_small_fwd', rowptr, colidx, val, X, H, Tc, Wg, Wc, dict(U=U, R=R, Cand=Cand, Hnew=Hnew, RH=RH), Zg, Zc)
            for name, t_ in (('Z0', Z0), ('Z0c', Z0c), ('Z1c', Z1c)):
                if t_ is not None:
                    self._f32('cell_small_fwd.' + name, t_, tuple(Zg.shape))
                    self._same_device(H, t_)
            for name, b_, n in (('bg', bg, 32), ('bc', bc, 16)):
                if b_ is not None:
                    self._f32('cell_small_fwd.' + name, b_, (n,))
        else:
            (B, N, Cc, _), cin, Kc = H.shape, X.shape[-1], Tc.shape[0]
        dense = int(is_full_pattern(colidx, N, N))
        g2, p2 = self._small_order3('cell_small_fwd', Tc, graph2, (Zg2, Zc2), Zg, Zc, dense, N)
        for phase in ((0,) if splits == 1 else self.SMALL_FWD_PHASES):
            self._launch('stc_cell_small_fwd_f32', H, rowptr.data_ptr(), colidx.data_ptr(), val.data_ptr(), N, val.numel(), dense, *g2, X.data_ptr(), cin,
                         H.data_ptr(), Tc.data_ptr(), Kc, Kc, Wg.data_ptr(), _ptr(bg), Wc.data_ptr(), _ptr(bc), U.data_ptr(), R.data_ptr(), Cand.data_ptr(),
                         Hnew.data_ptr(), RH.data_ptr(), Zg.data_ptr(), Zc.data_ptr(), *p2, _ptr(Z0), _ptr(Z0c), _ptr(Z1c), phase, splits, B, Cc,
                         nbytes=(4 * B * N * Cc * (cin + 16 * 8 + 2 * self.cell_small_zg_width(cin))) // (1 if splits == 1 else len(self.SMALL_FWD_PHASES)))

    # Launches of a split cell step (phase codes of stc_cell_small_*_f32; 5 = 1 + 2, 6 = 3 + 4, 7 = 2 + 3), CSR and dense graphs alike
    SMALL_FWD_PHASES = (5, 6)
    SMALL_BWD_PHASES = (1, 7, 4)

    @staticmethod
    def cell_small_splits(batch: int, rows: int = 0) -> int:
        """Workgroups per sample for a batch: 1 = one launch per cell step (a workgroup per sample); G > 1 = a few launches per step over
        G workgroups per sample, so that ~256 workgroups are in flight (rows: N * C of a sample -- a split wants at least a few tiles)."""
        forced = os.environ.get('STC_SMALL_SPLITS')                  # (probe: tools/gpu_ab.sh)
        if forced:
            return int(forced)
        g = max(1, min(8, 256 // max(1, batch)))
        while g > 1 and rows and rows < 48 * g:
            g //= 2
        return g

    def cell_small_bwd(self, rowptr, colidx, val, X, H, Tc, Wg, Wc, U, R, Cand, RH, Zg, Zc, dHnew, dX, accumulate_x, dH, accumulate_h,
                       dparams, has_bg, has_bc, checked=True, dZ1c=None, dZ1g=None, dYg=None, splits=1, dYc=None, graph2=None, Zg2=None, Zc2=None):
        """Autograd of ``cell_small_fwd`` in one launch (``stc_cell_small_bwd_f32``).  (rowptr, colidx, val): CSR of Gs (order 3: ``graph2`` =
        the CSR of 2 Gs^2 - I, ``Zg2`` / ``Zc2`` as the forward left them).  dX / dH may be
        None; ``accumulate_*``: add to what the buffer holds.  ``dparams`` (B * cell_small_param_rows, P >= cell_small_params):
        parameter-gradient partials (one row per sample and wave), ADDED to."""
        if checked:
            B, N, Cc, cin, Kc = self._small_shapes('cell_small_bwd', rowptr, colidx, val, X, H, Tc, Wg, Wc,
                                                   dict(U=U, R=R, Cand=Cand, RH=RH, dHnew=dHnew, **({} if dH is None else dict(dH=dH))), Zg, Zc)
            if dX is not None:
                self._f32('cell_small_bwd.dX', dX, tuple(X.shape))
            self._f32('cell_small_bwd.dparams', dparams)
            if dparams.dim() != 2 or dparams.shape[0] != B * splits * self.cell_small_param_rows or dparams.shape[1] < self.cell_small_params(Kc, Kc, cin):
                raise StcError(f'cell_small_bwd: dparams {tuple(dparams.shape)}, expected ({B * splits * self.cell_small_param_rows}, '
                               f'>= {self.cell_small_params(Kc, Kc, cin)})')
            for name, t_, shape in (('dZ1c', dZ1c, tuple(Zg.shape)), ('dZ1g', dZ1g, tuple(Zg.shape)), ('dYg', dYg, (B, N * Cc, 32)), ('dYc', dYc, (B, N * Cc, 16))):
                if t_ is not None:
                    self._f32('cell_small_bwd.' + name, t_, shape)
            self._same_device(H, dHnew, dparams, dZ1c, dZ1g, dYg, *([dX] if dX is not None else []))
        else:
            (B, N, Cc, _), cin, Kc = H.shape, X.shape[-1], Tc.shape[0]
        nbytes = self.lib.stc_cell_small_workspace_bytes(N, Cc, cin, B, Kc)
        ws = self._get_workspace(H.device, nbytes)
        dense = int(is_full_pattern(colidx, N, N))
        g2, p2 = self._small_order3('cell_small_bwd', Tc, graph2, (Zg2, Zc2), Zg, Zc, dense, N)
        phases = (0,) if splits == 1 else self.SMALL_BWD_PHASES
        for phase in phases:
            self._launch('stc_cell_small_bwd_f32', H, rowptr.data_ptr(), colidx.data_ptr(), val.data_ptr(), N, val.numel(), dense, *g2, X.data_ptr(), cin,
                         H.data_ptr(), Tc.data_ptr(), Kc, Kc, Wg.data_ptr(), Wc.data_ptr(), U.data_ptr(), R.data_ptr(), Cand.data_ptr(), RH.data_ptr(),
                         Zg.data_ptr(), Zc.data_ptr(), *p2, dHnew.data_ptr(), _ptr(dX), int(bool(accumulate_x)), _ptr(dH), int(bool(accumulate_h)),
                         dparams.data_ptr(), dparams.shape[1], int(bool(has_bg)), int(bool(has_bc)), _ptr(dZ1c), _ptr(dZ1g), _ptr(dYg), _ptr(dYc),
                         ws.data_ptr(), ws.numel(), phase, splits, B, Cc,
                         nbytes=(4 * B * N * Cc * (2 * cin + 16 * 9 + 5 * self.cell_small_zg_width(cin) + 64)) // len(phases))

    GRAD_CHUNKS = int(os.environ.get('STC_GRAD_CHUNKS', '96'))             # float64 partials of a graph-gradient product (x tile groups = workgroups; every partial is written and re-read)

    def _grad_operands(self, what, A, Bm, cell0, cell_step, n_sel, N):
        for name, t_ in (('A', A), ('B', Bm)):
            self._f32(f'{what}.{name}', t_)
            if t_.dim() != 4:
                raise StcError(f'{what}.{name}: expected (cells, batch, N*C, width), got {tuple(t_.shape)}')
        if A.shape[:3] != Bm.shape[:3] or A.shape[2] % N:
            raise StcError(f'{what}: operands {tuple(A.shape)} / {tuple(Bm.shape)} do not describe the same cells, samples and N = {N} nodes')
        if n_sel < 0 or cell0 < 0 or cell_step < 1 or (n_sel and cell0 + (n_sel - 1) * cell_step >= A.shape[0]):
            raise StcError(f'{what}: cells {cell0} + {cell_step} * [0, {n_sel}) outside the buffer of {A.shape[0]} cells')
        self._same_device(A, Bm)
        return A.shape[1], A.shape[2] // N

    def grad_partials(self, like, total, chunks=None):
        """(chunks, total) float64 buffer for the blocks of several graph-gradient products side by side (``graph_grad`` / ``mix_grad`` with
        ``into=``): ONE ``sum(0)`` then adds the partials of all of them."""
        return torch.empty(self.GRAD_CHUNKS if chunks is None else chunks, total, dtype=torch.float64, device=like.device)

    def _grad_block(self, what, into, size):
        part, offset = into
        if part.dtype != torch.float64 or part.dim() != 2 or not part.is_contiguous() or not part.is_cuda or offset < 0 or offset + size > part.shape[1]:
            raise StcError(f'{what}: into = (contiguous float64 (chunks, total) ROCm buffer, column offset); block of {size} at {offset} in {tuple(part.shape)}')
        return part, part.data_ptr() + 8 * offset, part.shape[0], part.shape[1]

    def graph_grad(self, A, Bm, cell0, cell_step, n_sel, N, into=None):
        """(N, N) float64:  sum over the selected cells and samples of  A_g . B_g^T  with A, B (cells, batch, N*C, width) read as (N, C*width)
        per plane -- the dGs^T piece of a learned dense graph (``stc_graph_grad_f32``).  ``into`` = (partial buffer, column offset): the
        partials go to that block of a shared buffer (``grad_partials``) and nothing is returned -- the caller sums the buffer once."""
        batch, Cc = self._grad_operands('graph_grad', A, Bm, cell0, cell_step, n_sel, N)
        if A.shape[3] != Bm.shape[3]:
            raise StcError(f'graph_grad: widths {A.shape[3]} / {Bm.shape[3]} differ')
        if into is not None:
            part, ptr, chunks, stride = self._grad_block('graph_grad', into, N * N)
            self._same_device(A, part)
            self._launch('stc_graph_grad_f32', A, A.data_ptr(), Bm.data_ptr(), ptr, chunks, cell0, cell_step, n_sel, batch, N, Cc * A.shape[3], stride)
            return None
        chunks = max(1, min(256, n_sel * batch))          # (an N x N float64 partial is small: one workgroup per ~2 planes)
        part = torch.empty(chunks, N, N, dtype=torch.float64, device=A.device)
        self._launch('stc_graph_grad_f32', A, A.data_ptr(), Bm.data_ptr(), part.data_ptr(), chunks, cell0, cell_step, n_sel, batch, N, Cc * A.shape[3], 0)
        return part.sum(0)

    def mix_grad(self, A, Bm, cell0, cell_step, n_sel, N, into=None):
        """(C*wa, C*wb) float64:  sum over the selected cells and samples of  A_g^T . B_g  (contraction over the N nodes) -- Q = Z^T . dY of one
        slab of one convolution (``stc_mix_grad_f32``).  ``into``: as ``graph_grad``."""
        batch, Cc = self._grad_operands('mix_grad', A, Bm, cell0, cell_step, n_sel, N)
        Fa, Fb = Cc * A.shape[3], Cc * Bm.shape[3]
        if into is not None:
            part, ptr, chunks, stride = self._grad_block('mix_grad', into, Fa * Fb)
            self._same_device(A, part)
            self._launch('stc_mix_grad_f32', A, A.data_ptr(), Bm.data_ptr(), ptr, chunks, cell0, cell_step, n_sel, batch, N, Fa, Fb, stride)
            return None
        chunks = max(1, min(self.GRAD_CHUNKS, n_sel * batch))
        part = torch.empty(chunks, Fa, Fb, dtype=torch.float64, device=A.device)
        self._launch('stc_mix_grad_f32', A, A.data_ptr(), Bm.data_ptr(), part.data_ptr(), chunks, cell0, cell_step, n_sel, batch, N, Fa, Fb, 0)
        return part.sum(0)

    # ---- MixedFusion of the learned graph generator (reference STC_GNN.py:246-261) ----------------------------------
    def mixed_fusion_supported(self, D) -> bool:
        return D >= 4 and D % 4 == 0

    def mixed_fusion_fwd(self, WA, bA, WP, bP, A, P):
        """(gate, G) with gate = sigmoid(W_A vec(A) + b_A + W_P vec(P) + b_P), G = gate * A + (1 - gate) * P; A, P (n, n) (``stc_mixed_fusion_fwd_f32``)."""
        D = A.numel()
        for name, t, shape in (('WA', WA, (D, D)), ('WP', WP, (D, D)), ('bA', bA, (D,)), ('bP', bP, (D,))):
            self._f32('mixed_fusion.' + name, t, shape)
        self._f32('mixed_fusion.A', A)
        self._f32('mixed_fusion.P', P, tuple(A.shape))
        self._same_device(WA, bA, WP, bP, A, P)
        gate, G = torch.empty_like(A), torch.empty_like(A)
        self._launch('stc_mixed_fusion_fwd_f32', A, _ptr(WA), _ptr(bA), _ptr(WP), _ptr(bP), _ptr(A), _ptr(P), _ptr(gate), _ptr(G), D, nbytes=8 * D * D)
        return gate, G

    def mixed_fusion_bwd(self, WA, WP, A, P, gate, dG, want_dA, want_dW=True):
        """(dW_A, dW_P, db, dP, dA or None) from dG (``stc_mixed_fusion_bwd_f32``); db is the gradient of both biases; ``want_dW=False``
        (frozen weights): the two (D, D) gradients are neither allocated nor written (None, None)."""
        D = A.numel()
        for name, t in (('gate', gate), ('dG', dG), ('P', P)):
            self._f32('mixed_fusion.' + name, t, tuple(A.shape))
        self._same_device(WA, WP, A, P, gate, dG)
        dWA, dWP = (torch.empty_like(WA), torch.empty_like(WP)) if want_dW else (None, None)
        db, dP = torch.empty(D, dtype=torch.float32, device=A.device), torch.empty_like(A)
        dA = torch.empty_like(A) if want_dA else None
        ws = torch.empty(self.lib.stc_mixed_fusion_workspace_bytes(D, int(want_dA)) // 4, dtype=torch.float32, device=A.device)
        self._launch('stc_mixed_fusion_bwd_f32', A, _ptr(WA), _ptr(WP), _ptr(A), _ptr(P), _ptr(gate), _ptr(dG), _ptr(dWA), _ptr(dWP), _ptr(db), _ptr(dP), _ptr(dA),
                     _ptr(ws), ws.numel() * 4, D, nbytes=(4 + (8 if want_dW else 0) + (4 if want_dA else 0)) * D * D)
        return dWA, dWP, db, dP, dA

    # ---- front end of the learned graph generator (reference STC_GNN.py:229-232, 237-240) --------------------------
    @staticmethod
    def _mgp_strides(X, rows_axis):
        """(K, R, F, k_stride, r_stride, f_stride) of a contiguous window X (B, T, N, C) read as x[k][r][f]: rows_axis 2 = the spatial branch (rows =
        nodes, features = categories), 3 = the category branch (the transposed window of :236)."""
        B, T, N, Cc = X.shape
        return (B * T, N, Cc, N * Cc, Cc, 1) if rows_axis == 2 else (B * T, Cc, N, N * Cc, 1, Cc)

    def mgp_uv_fwd(self, X, rows_axis, Wu, Wv, alpha):
        """(U, V), each (R, K, h): tanh(alpha x Wu), tanh(alpha x Wv) with the (sample, time) slices as the middle axis (``stc_mgp_uv_fwd_f32``)."""
        self._f32('mgp.X', X)
        K, R, F, ks, rs, fs = self._mgp_strides(X, rows_axis)
        h = Wu.shape[1]
        self._f32('mgp.Wu', Wu, (F, h))
        self._f32('mgp.Wv', Wv, (F, h))
        self._same_device(X, Wu, Wv)
        U, V = torch.empty(R, K, h, dtype=torch.float32, device=X.device), torch.empty(R, K, h, dtype=torch.float32, device=X.device)
        self._launch('stc_mgp_uv_fwd_f32', X, _ptr(X), ks, rs, fs, _ptr(Wu), _ptr(Wv), float(alpha), _ptr(U), _ptr(V), K, R, F, h)
        return U, V

    def mgp_uv_bwd(self, X, rows_axis, U, V, dU, dV, alpha):
        """(dWu, dWv) (F, h) from the gradients of U and V (``stc_mgp_uv_bwd_f32`` + a fixed-order sum over the slices)."""
        K, R, F, ks, rs, fs = self._mgp_strides(X, rows_axis)
        h = U.shape[2]
        for name, t in (('U', U), ('V', V), ('dU', dU), ('dV', dV)):
            self._f32('mgp.' + name, t, (R, K, h))
        self._same_device(X, U, V, dU, dV)
        part = torch.empty(K, 2, F, h, dtype=torch.float32, device=X.device)
        self._launch('stc_mgp_uv_bwd_f32', X, _ptr(X), ks, rs, fs, _ptr(U), _ptr(V), _ptr(dU), _ptr(dV), float(alpha), _ptr(part), K, R, F, h)
        dW = part.sum(0)
        return dW[0], dW[1]

    def mgp_softmax_fwd(self, P):
        """softmax(relu(P - P^T), -1) of a square P (``stc_mgp_softmax_fwd_f32``)."""
        self._f32('mgp.P', P)
        if P.dim() != 2 or P.shape[0] != P.shape[1]:
            raise StcError(f'mgp.P: square matrix expected, got {tuple(P.shape)}')
        Ps = torch.empty_like(P)
        self._launch('stc_mgp_softmax_fwd_f32', P, _ptr(P), _ptr(Ps), P.shape[0])
        return Ps

    def mgp_softmax_bwd(self, P, Ps, dPs):
        """dP from dPs (``stc_mgp_softmax_bwd_f32``)."""
        for name, t in (('Ps', Ps), ('dPs', dPs)):
            self._f32('mgp.' + name, t, tuple(P.shape))
        self._same_device(P, Ps, dPs)
        R = P.shape[0]
        rowdot, dP = torch.empty(R, dtype=torch.float32, device=P.device), torch.empty_like(P)
        self._launch('stc_mgp_softmax_bwd_f32', P, _ptr(P), _ptr(Ps), _ptr(dPs), _ptr(rowdot), _ptr(dP), R)
        return dP

    def adam(self, p, g, m, v, step, lr, beta1, beta2, eps, weight_decay):
        """One Adam update of a large fp32 parameter in place (``stc_adam_f32``: torch.optim.Adam's arithmetic with L2 weight decay); ``step``: a
        one-element float32 DEVICE tensor holding the count of this update (already incremented)."""
        n = p.numel()
        for name, t in (('p', p), ('g', g), ('m', m), ('v', v)):
            self._f32('adam.' + name, t, tuple(p.shape))
        self._f32('adam.step', step)
        if step.numel() != 1:
            raise StcError(f'adam.step: one element, got {tuple(step.shape)}')
        self._same_device(p, g, m, v, step)
        self._launch('stc_adam_f32', p, _ptr(p), _ptr(g), _ptr(m), _ptr(v), n, _ptr(step), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                     nbytes=28 * n)

    # ---- planar cell convolutions of Chebyshev order K = 3 ------------------------------------------------
    def cell_planar_k_supported(self, K, Cc, h) -> bool:
        return bool(self.lib.stc_cell_planar_k_supported(K, Cc, h))

    def _planes_k(self, what, Zx, Zh, Tc, W, Ho):
        """Zx / Zh: K planes each, T_n(S) of the X-side / H-side plane: Zh[n] (R, C, h); Zx[n] (R, C, cin), cin = h or 1..4."""
        K = len(Zh)
        if len(Zx) != K or K < 2:
            raise StcError(f'{what}: {len(Zx)} X-side planes for {K} H-side planes')
        R, Cc, h = Zh[0].shape
        cin = Zx[0].shape[-1]
        if not (cin == h or 1 <= cin <= 4):
            raise StcError(f'{what}: input plane width {cin} must be {h} or 1..4')
        for n in range(K):
            self._f32(f'{what}.Zh[{n}]', Zh[n], (R, Cc, h))
            self._f32(f'{what}.Zx[{n}]', Zx[n], (R, Cc, cin))
        self._f32(what + '.Tc', Tc, (K, Cc, Cc))
        self._f32(what + '.W', W, (K * K * (cin + h), Ho))
        return K, R, Cc, h, cin

    def cell_gates_fwd_planar_k(self, Zx, Zh, Tc, W, bias, U, Rg, RH, act_amax=None):
        K, R, Cc, h, cin = self._planes_k('planar_k gates', Zx, Zh, Tc, W, 2 * Zh[0].shape[-1])
        if bias is not None:
            self._f32('planar_k.bias', bias, (2 * h,))
        for name, t in (('U', U), ('Rg', Rg), ('RH', RH)):
            self._f32('planar_k.' + name, t, (R, Cc, h))
        self._same_device(*Zx, *Zh, Tc, W, bias, U, Rg, RH)
        self._launch('stc_cell_gates_fwd_planar_k_f32', U, self._ptr_array(Zx), self._ptr_array(Zh), K, _ptr(Tc), _ptr(W), _ptr(bias),
                     _ptr(U), _ptr(Rg), _ptr(RH), self.operand_format, self._act_amax('planar_k', act_amax, 2 * K, U), R, Cc, cin + h, h,
                     nbytes=4 * R * Cc * (K * (cin + h) + 3 * h))

    def cell_cand_fwd_planar_k(self, Zx, Zh, Tc, W, bias, U, H, Cand, Hnew, act_amax=None):
        """Candidate convolution on [X | R*H] (Zh = the T_n(S) planes of R*H) + tanh + GRU blend: Cand, Hnew."""
        K, R, Cc, h, cin = self._planes_k('planar_k cand', Zx, Zh, Tc, W, Zh[0].shape[-1])
        if bias is not None:
            self._f32('planar_k.bias', bias, (h,))
        for name, t in (('U', U), ('H', H), ('Cand', Cand), ('Hnew', Hnew)):
            self._f32('planar_k.' + name, t, (R, Cc, h))
        self._same_device(*Zx, *Zh, Tc, W, bias, U, H, Cand, Hnew)
        self._launch('stc_cell_cand_fwd_planar_k_f32', U, self._ptr_array(Zx), self._ptr_array(Zh), K, _ptr(Tc), _ptr(W), _ptr(bias),
                     _ptr(U), _ptr(H), _ptr(Cand), _ptr(Hnew), self.operand_format, self._act_amax('planar_k', act_amax, 2 * K, U), R, Cc, cin + h, h,
                     nbytes=4 * R * Cc * (K * (cin + h) + 4 * h))

    def _grad_planes_k(self, what, dZx, dZh, K, R, Cc, h, cin):
        if len(dZh) != K or len(dZx) != K:
            raise StcError(f'{what}: need {K} gradient planes per side')
        for n in range(K):
            self._f32(f'{what}.dZh[{n}]', dZh[n], (R, Cc, h))
            if dZx[n] is None and cin != h:
                continue                                              # a narrow input plane needs no gradient
            self._f32(f'{what}.dZx[{n}]', dZx[n], (R, Cc, cin))
        return (_p * K)(*[0 if z is None else z.data_ptr() for z in dZx]), (_p * K)(*[z.data_ptr() for z in dZh])

    def cell_gates_bwd_planar_k(self, Zx, Zh, Tc, W, dRH, Cand, U, Rg, dHnew, dZx, dZh, dW, db, dH, accumulate_x=False, act_amax=None):
        """``accumulate_x``: the X-side planes ``dZx`` hold the candidate's gradients of the same planes; the gates' are added to them
        (wide input, ``dH=None`` only)."""
        K, R, Cc, h, cin = self._planes_k('planar_k gates bwd', Zx, Zh, Tc, W, 2 * Zh[0].shape[-1])
        if accumulate_x and (dH is not None or cin != h):
            raise StcError('planar_k gates bwd: accumulate_x goes with a 16-wide input and dH=None')
        for name, t in (('dRH', dRH), ('Cand', Cand), ('U', U), ('Rg', Rg), ('dHnew', dHnew)) + ((('dH', dH),) if dH is not None else ()):
            self._f32('planar_k.' + name, t, (R, Cc, h))    # dH None: the kernel folds the state's share into dZh[0]
        zx, zh = self._grad_planes_k('planar_k gates bwd', dZx, dZh, K, R, Cc, h, cin)
        self._f32('planar_k.dW', dW, tuple(W.shape))
        if db is not None:
            self._f32('planar_k.db', db, (2 * h,))
        self._same_device(*Zx, *Zh, Tc, W, dRH, Cand, U, Rg, dHnew, *dZx, *dZh, dW, db, dH)
        ws = self._get_workspace(U.device, self.lib.stc_bdg_node_bwd_workspace_bytes(K, K, Cc, 2 * h, 2 * h, 0))
        self._launch('stc_cell_gates_bwd_planar_k_f32', U, self._ptr_array(Zx), self._ptr_array(Zh), K, _ptr(Tc), _ptr(W), _ptr(dRH), _ptr(Cand),
                     _ptr(U), _ptr(Rg), _ptr(dHnew), zx, zh, _ptr(dW), _ptr(db), _ptr(dH), int(bool(accumulate_x)),
                     self.operand_format, self._act_amax('planar_k', act_amax, 2 * K, U), _ptr(ws), ws.numel(), R, Cc, cin + h, h,
                     nbytes=4 * R * Cc * (K * (cin + h) + 5 * h + K * h + (K * h * (2 if accumulate_x else 1) if cin == h else 0) + (h if dH is not None else 0)))

    def cell_cand_bwd_planar_k(self, Zx, Zh, Tc, W, dHnew, U, Cand, dZx, dZh, dW, db, act_amax=None):
        K, R, Cc, h, cin = self._planes_k('planar_k cand bwd', Zx, Zh, Tc, W, Zh[0].shape[-1])
        for name, t in (('dHnew', dHnew), ('U', U), ('Cand', Cand)):
            self._f32('planar_k.' + name, t, (R, Cc, h))
        zx, zh = self._grad_planes_k('planar_k cand bwd', dZx, dZh, K, R, Cc, h, cin)
        self._f32('planar_k.dW', dW, tuple(W.shape))
        if db is not None:
            self._f32('planar_k.db', db, (h,))
        self._same_device(*Zx, *Zh, Tc, W, dHnew, U, Cand, *dZx, *dZh, dW, db)
        ws = self._get_workspace(U.device, self.lib.stc_bdg_node_bwd_workspace_bytes(K, K, Cc, 2 * h, 2 * h, 0))
        self._launch('stc_cell_cand_bwd_planar_k_f32', U, self._ptr_array(Zx), self._ptr_array(Zh), K, _ptr(Tc), _ptr(W), _ptr(dHnew), _ptr(U), _ptr(Cand),
                     zx, zh, _ptr(dW), _ptr(db), self.operand_format, self._act_amax('planar_k', act_amax, 2 * K, U),
                     _ptr(ws), ws.numel(), R, Cc, cin + h, h,
                     nbytes=4 * R * Cc * (K * (cin + h) + 3 * h + K * h + (K * h if cin == h else 0)))

    # ---- fused cell convolutions ----------------------------------------------------------
    def cell_fused_supported(self, Ks, Kc, Cc, L, h) -> bool:
        return bool(self.lib.stc_cell_fused_supported(Ks, Kc, Cc, L, h))

    def cell_gates_fwd(self, Zs, Tc, W, bias, H, U, Rg, CandIn):
        Ks, Kc, R, Cc, L, Lw, Ho = self._node_shapes(Zs, Tc, W)
        h = H.shape[-1]
        cin = Lw - h
        if Ho != 2 * h or cin < 0:
            raise StcError(f'cell_gates: W gives Ho={Ho}, Lw={Lw} for hidden {h}')
        if bias is not None:
            self._f32('cell.bias', bias, (Ho,))
        for name, t in (('H', H), ('U', U), ('Rg', Rg)):
            self._f32('cell.' + name, t, (R, Cc, h))
        self._f32('cell.CandIn', CandIn, (R, Cc, L))
        self._same_device(*Zs, Tc, W, bias, H, U, Rg, CandIn)
        self._launch('stc_cell_gates_fwd_f32', H, self._ptr_array(Zs), Ks, _ptr(Tc), Kc, _ptr(W), _ptr(bias), _ptr(H),
                     _ptr(U), _ptr(Rg), _ptr(CandIn), R, Cc, L, Lw, h, cin)

    def cell_gates_bwd(self, Zs, Tc, W, dCandIn, dU, H, U, Rg, dH_in, dZs, dW, db, dXt, dH, dH_in_scaled=False, Cand=None):
        """Gate backward as the prologue of the gates convolution's node backward (stc_cell_gates_bwd_f32).
        ``dXt`` may be None (not written); ``dH_in_scaled``: dH_in enters times (1 - U); with ``Cand`` (and dU None)
        dH_in is dHnew and dU = dHnew * (Cand - H) is formed inside."""
        if (dU is None) == (Cand is None):
            raise StcError('cell_gates_bwd: give either dU or Cand')
        Ks, Kc, R, Cc, L, Lw, Ho = self._node_shapes(Zs, Tc, W)
        h = H.shape[-1]
        cin = Lw - h
        if Ho != 2 * h or cin < 0 or len(dZs) != Ks:
            raise StcError(f'cell_gates_bwd: W gives Ho={Ho}, Lw={Lw} for hidden {h}; {len(dZs)} gradient slabs')
        self._f32('cell.dCandIn', dCandIn, (R, Cc, L))
        for name, t in (('dU', dU), ('Cand', Cand), ('H', H), ('U', U), ('Rg', Rg), ('dH', dH), ('dH_in', dH_in)):
            if t is not None:
                self._f32('cell.' + name, t, (R, Cc, h))
        if dXt is not None:
            self._f32('cell.dXt', dXt, (R, Cc, cin))
        for i, z in enumerate(dZs):
            self._f32(f'cell.dZ[{i}]', z, (R, Cc, L))
        self._f32('cell.dW', dW, (Ks * Kc * Lw, Ho))
        if db is not None:
            self._f32('cell.db', db, (Ho,))
        self._same_device(*Zs, Tc, W, dCandIn, dU, H, U, Rg, dH_in, Cand, *dZs, dW, db, dXt, dH)
        ws = self._get_workspace(H.device, self.lib.stc_bdg_node_bwd_workspace_bytes(Ks, Kc, Cc, L, Ho, 0))
        self._launch('stc_cell_gates_bwd_f32', H, self._ptr_array(Zs), Ks, _ptr(Tc), Kc, _ptr(W), _ptr(dCandIn), _ptr(dU), _ptr(H),
                     _ptr(U), _ptr(Rg), _ptr(Cand), _ptr(dH_in), int(bool(dH_in_scaled)), self._ptr_array(dZs), _ptr(dW), _ptr(db), _ptr(dXt), _ptr(dH),
                     _ptr(ws), ws.numel(), R, Cc, L, Lw, h, cin)

    def cell_cand_bwd(self, Zs, Tc, W, dHnew, U, Cand, dZs, dW, db):
        """Blend backward as the prologue of the candidate convolution's node backward (stc_cell_cand_bwd_f32)."""
        Ks, Kc, R, Cc, L, Lw, Ho = self._node_shapes(Zs, Tc, W)
        h = U.shape[-1]
        if Ho != h or len(dZs) != Ks:
            raise StcError(f'cell_cand_bwd: W gives Ho={Ho} for hidden {h}; {len(dZs)} gradient slabs')
        for name, t in (('dHnew', dHnew), ('U', U), ('Cand', Cand)):
            self._f32('cell.' + name, t, (R, Cc, h))
        for i, z in enumerate(dZs):
            self._f32(f'cell.dZ[{i}]', z, (R, Cc, L))
        self._f32('cell.dW', dW, (Ks * Kc * Lw, Ho))
        if db is not None:
            self._f32('cell.db', db, (Ho,))
        self._same_device(*Zs, Tc, W, dHnew, U, Cand, *dZs, dW, db)
        ws = self._get_workspace(U.device, self.lib.stc_bdg_node_bwd_workspace_bytes(Ks, Kc, Cc, L, Ho, 0))
        self._launch('stc_cell_cand_bwd_f32', U, self._ptr_array(Zs), Ks, _ptr(Tc), Kc, _ptr(W), _ptr(dHnew), _ptr(U), _ptr(Cand),
                     self._ptr_array(dZs), _ptr(dW), _ptr(db), _ptr(ws), ws.numel(), R, Cc, L, Lw, h)

    def cell_blend_fwd(self, Zs, Tc, W, bias, U, H, Cand, Hnew, copies=(), side=None):
        """``copies``: up to two (buffer, column offset) pairs -- the new state is also written into columns
        [off, off + h) of those (rows, C, ld) buffers (the input rows of the cells that consume it).  ``side``: a
        (rows, C, cin) tensor completing the FIRST copy's row: its columns [0, cin) are filled from it (cin == off) and
        its pad columns zeroed."""
        Ks, Kc, R, Cc, L, Lw, Ho = self._node_shapes(Zs, Tc, W)
        h = H.shape[-1]
        if Ho != h:
            raise StcError(f'cell_blend: W gives Ho={Ho} for hidden {h}')
        if bias is not None:
            self._f32('cell.bias', bias, (Ho,))
        for name, t in (('U', U), ('H', H), ('Cand', Cand), ('Hnew', Hnew)):
            self._f32('cell.' + name, t, (R, Cc, h))
        if len(copies) > 2 or (side is not None and not copies):
            raise StcError('cell_blend: at most two state copies; side needs a first copy')
        cp = []
        for i, (buf, off) in enumerate(copies):
            self._f32(f'cell.copy{i}', buf)
            if buf.dim() != 3 or buf.shape[:2] != (R, Cc) or off < 0 or off + h > buf.shape[-1]:
                raise StcError(f'cell_blend: copy{i} of shape {tuple(buf.shape)} cannot take columns [{off}, {off + h}) of {R} x {Cc} rows')
            cp.append((buf, buf.shape[-1], off))
        while len(cp) < 2:
            cp.append((None, 0, 0))
        side_cin = 0
        if side is not None:
            side_cin = side.shape[-1]
            self._f32('cell.side', side, (R, Cc, side_cin))
            if side_cin != cp[0][2]:
                raise StcError(f'cell_blend: side width {side_cin} must equal the first copy\'s column offset {cp[0][2]}')
        self._same_device(*Zs, Tc, W, bias, U, H, Cand, Hnew, cp[0][0], cp[1][0], side)
        self._launch('stc_cell_blend_fwd_f32', H, self._ptr_array(Zs), Ks, _ptr(Tc), Kc, _ptr(W), _ptr(bias), _ptr(U), _ptr(H),
                     _ptr(Cand), _ptr(Hnew), _ptr(cp[0][0]), cp[0][1], cp[0][2], _ptr(side), side_cin, _ptr(cp[1][0]), cp[1][1], cp[1][2],
                     R, Cc, L, Lw, h)

    # ---- GRU gate math -------------------------------------------------------------------
    def gru_gates_fwd(self, G, Xt, H, U, Rg, CandIn):
        rows, h = H.shape[:-1].numel(), H.shape[-1]
        cin = Xt.shape[-1]
        pad = CandIn.shape[-1] - cin - h
        if pad < 0:
            raise StcError(f'gates.CandIn: width {CandIn.shape[-1]} < cin + h = {cin + h}')
        for name, t, w in (('G', G, 2 * h), ('Xt', Xt, cin), ('H', H, h), ('U', U, h), ('Rg', Rg, h),
                           ('CandIn', CandIn, cin + h + pad)):
            self._f32('gates.' + name, t)
            if t.shape[-1] != w or t.numel() != rows * w:
                raise StcError(f'gates.{name}: shape {tuple(t.shape)} does not match rows={rows}, width={w}')
        self._launch('stc_gru_gates_fwd_f32', H, _ptr(G), _ptr(Xt), _ptr(H), _ptr(U), _ptr(Rg), _ptr(CandIn), rows, cin, h, pad)

    def gru_gates_bwd(self, dCandIn, dU, H, U, Rg, dG, dXt, dH, dH_in=None):
        rows, h = H.shape[:-1].numel(), H.shape[-1]
        cin = dXt.shape[-1]
        pad = dCandIn.shape[-1] - cin - h
        if pad < 0:
            raise StcError(f'gates_bwd.dCandIn: width {dCandIn.shape[-1]} < cin + h = {cin + h}')
        for name, t, w in (('dCandIn', dCandIn, cin + h + pad), ('dU', dU, h), ('H', H, h), ('U', U, h), ('Rg', Rg, h),
                           ('dG', dG, 2 * h), ('dXt', dXt, cin), ('dH', dH, h)) + ((('dH_in', dH_in, h),) if dH_in is not None else ()):
            self._f32('gates_bwd.' + name, t)
            if t.shape[-1] != w or t.numel() != rows * w:
                raise StcError(f'gates_bwd.{name}: shape {tuple(t.shape)} does not match rows={rows}, width={w}')
        self._launch('stc_gru_gates_bwd_f32', H, _ptr(dCandIn), _ptr(dU), _ptr(H), _ptr(U), _ptr(Rg), _ptr(dH_in), _ptr(dG), _ptr(dXt), _ptr(dH), rows, cin, h, pad)

    def _same_numel(self, what, *ts):
        n = ts[0].numel()
        for i, t in enumerate(ts):
            self._f32(f'{what}[{i}]', t)
            if t.numel() != n:
                raise StcError(f'{what}: operand {i} has {t.numel()} elements, expected {n}')
        return n

    def gru_blend_fwd(self, Cpre, U, H, Cand, Hnew):
        n = self._same_numel('blend', Cpre, U, H, Cand, Hnew)
        self._launch('stc_gru_blend_fwd_f32', H, _ptr(Cpre), _ptr(U), _ptr(H), _ptr(Cand), _ptr(Hnew), n)

    def gru_blend_bwd(self, dHnew, U, H, Cand, dCpre, dU, dH):
        """``dU`` / ``dH`` may be None (those products are then formed by the consumer); ``H`` may be None when dU is."""
        n = self._same_numel('blend_bwd', dHnew, U, Cand, dCpre, *(t for t in (H, dU, dH) if t is not None))
        if dU is not None and H is None:
            raise StcError('blend_bwd: dU needs H')
        self._launch('stc_gru_blend_bwd_f32', U, _ptr(dHnew), _ptr(U), _ptr(H), _ptr(Cand), _ptr(dCpre), _ptr(dU), _ptr(dH), n)

    # ---- output head -----------------------------------------------------------------------
    def head_fwd(self, H, w, b, y):
        h = H.shape[-1]
        rows = H.shape[:-1].numel()
        if H.dtype == torch.bfloat16:                       # bf16 state rows (hidden 16); y stays fp32
            self._bf16('head.H', H)
            self._f32('head.w', w, (h,))
            self._f32('head.b', b, (1,))
            self._f32('head.y', y, tuple(H.shape[:-1]))
            self._launch('stc_head_fwd_bf16', H, _ptr(H), _ptr(w), _ptr(b), _ptr(y), rows, h)
            return
        self._f32('head.H', H)
        self._f32('head.w', w, (h,))
        self._f32('head.b', b, (1,))
        self._f32('head.y', y, tuple(H.shape[:-1]))
        self._launch('stc_head_fwd_f32', H, _ptr(H), _ptr(w), _ptr(b), _ptr(y), rows, h)

    def head_bwd(self, H, w, y, dy, dH, dwb):
        h = H.shape[-1]
        rows = H.shape[:-1].numel()
        if H.dtype == torch.bfloat16:
            self._bf16('head.H', H)
            self._bf16('head.dH', dH, tuple(H.shape))
            self._f32('head.w', w, (h,))
            self._f32('head.y', y, tuple(H.shape[:-1]))
            self._f32('head.dy', dy, tuple(H.shape[:-1]))
            self._f32('head.dwb', dwb, (h + 1,))
            ws = self._get_workspace(H.device, self.lib.stc_head_bwd_workspace_bytes(h))
            self._launch('stc_head_bwd_bf16', H, _ptr(H), _ptr(w), _ptr(y), _ptr(dy), _ptr(dH), _ptr(dwb), _ptr(ws), ws.numel(), rows, h)
            return
        self._f32('head.H', H)
        self._f32('head.w', w, (h,))
        self._f32('head.y', y, tuple(H.shape[:-1]))
        self._f32('head.dy', dy, tuple(H.shape[:-1]))
        self._f32('head.dH', dH, tuple(H.shape))
        self._f32('head.dwb', dwb, (h + 1,))
        ws = self._get_workspace(H.device, self.lib.stc_head_bwd_workspace_bytes(h))
        self._launch('stc_head_bwd_f32', H, _ptr(H), _ptr(w), _ptr(y), _ptr(dy), _ptr(dH), _ptr(dwb), _ptr(ws), ws.numel(), rows, h)

    # ---- helpers --------------------------------------------------------------------------
    def axpy(self, a, x, y):
        n = self._same_numel('axpy', x, y)
        self._launch('stc_axpy_f32', x, float(a), _ptr(x), _ptr(y), n)

    def _cat_shapes(self, what, A, Bm, whole):
        a, b = A.shape[-1], Bm.shape[-1]
        rows = whole.shape[:-1].numel()
        pad = whole.shape[-1] - a - b
        if pad < 0:
            raise StcError(f'{what}: joined width {whole.shape[-1]} < {a} + {b}')
        for name, t, w in (('A', A, a), ('B', Bm, b), ('whole', whole, a + b + pad)):
            self._f32(f'{what}.{name}', t)
            if t.shape[-1] != w or t.numel() != rows * w:
                raise StcError(f'{what}.{name}: shape {tuple(t.shape)} does not match rows={rows}, width={w}')
        return rows, a, b, pad

    def concat2(self, A, Bm, out):
        rows, a, b, pad = self._cat_shapes('concat2', A, Bm, out)
        self._launch('stc_concat2_f32', out, _ptr(A), _ptr(Bm), _ptr(out), rows, a, b, pad)

    def split2(self, src, A, Bm, addA=None, addB=None, addA_ld=0, addA2=None, addB2=None):
        """``addA_ld`` > 0: ``addA`` is a (rows, addA_ld) buffer whose first ``a`` columns are added (read in place).
        ``addA2`` / ``addB2``: one more addend each (e.g. what the halves are already owed from another consumer)."""
        rows, a, b, pad = self._cat_shapes('split2', A, Bm, src)
        if addA is not None:
            self._f32('split2.addA', addA, A.shape if not addA_ld else A.shape[:-1] + (addA_ld,))
            if addA_ld and addA_ld < a:
                raise StcError(f'split2: addA_ld={addA_ld} is smaller than the width {a}')
        for name, t, like in (('addB', addB, Bm), ('addA2', addA2, A), ('addB2', addB2, Bm)):
            if t is not None:
                self._f32('split2.' + name, t, like.shape)
        self._launch('stc_split2_f32', src, _ptr(src), _ptr(addA), _ptr(addB), _ptr(A), _ptr(Bm), rows, a, b, pad,
                     int(addA_ld if addA is not None else 0), _ptr(addA2), _ptr(addB2))


class _Bf16Planar:
    """bf16-plane counterpart of the planar-cell methods of ``HipKernels`` (same names and argument meaning, so that the
    cell-graph executor runs unchanged): state / gate / gradient planes bfloat16 (R, C, 16), weights and their gradients fp32.
    Only what an all-planar schedule needs exists here; interleaved rows and state copies are fp32-path features."""

    name = 'hip-gfx950-bf16'
    folds_dH = True          # cell_gates_bwd_planar(dH=None) adds the prologue's share of the previous state into the H plane's gradient

    def __init__(self, base: HipKernels):
        self.b = base

    def _pl(self, name, t, shape):
        return self.b._bf16(name, t, shape)

    def cell_fused_supported(self, Ks, Kc, Cc, L, h) -> bool:
        return self.cell_planar_supported(Ks, Kc, Cc, h)

    def cell_planar_supported(self, Ks, Kc, Cc, h) -> bool:
        return bool(self.b.lib.stc_cell_planar_bf16_supported(Ks, Kc, Cc, h))

    def node_post_supported(self, Ks, Kc, Cc, L, Ho) -> bool:
        return self.cell_planar_supported(Ks, Kc, Cc, Ho) and L in (20, 32)

    def cell_planar_post_fused(self, Cc) -> bool:
        return True

    def csr_spmm(self, rowptr, colidx, val, n_rows, n_cols, X, Y0, Y, alpha, beta, plan=None):
        self.b.csr_spmm_bf16(rowptr, colidx, val, n_rows, n_cols, X, Y0, Y, alpha, beta, plan=plan)

    def _planes(self, X, H, SX, SH):
        R, Cc, h = H.shape
        cin = X.shape[-1]
        if h != 16 or not (cin == h or 1 <= cin <= 4):
            raise StcError(f'planar bf16: hidden {h} / input plane width {cin} (16, and 16 or 1..4)')
        for name, t in (('H', H), ('SH', SH)):
            self._pl('planar.' + name, t, (R, Cc, h))
        for name, t in (('X', X), ('SX', SX)):
            self._pl('planar.' + name, t, (R, Cc, cin))
        return R, Cc, h, cin

    def cell_gates_fwd_planar(self, X, H, SX, SH, Tc, W, bias, U, Rg, RH, post=None):
        b = self.b
        R, Cc, h, cin = self._planes(X, H, SX, SH)
        b._f32('planar.Tc', Tc, (2, Cc, Cc))
        b._f32('planar.W', W, (4 * (cin + h), 2 * h))
        if bias is not None:
            b._f32('planar.bias', bias, (2 * h,))
        if RH is None and post is None:
            raise StcError('planar gates forward (bf16): the R*H plane is optional only with the fused candidate projection (post=)')
        for name, t in (('U', U), ('Rg', Rg)) + ((('RH', RH),) if RH is not None else ()):
            self._pl('planar.' + name, t, (R, Cc, h))
        Wc = bc = A = Bm = None
        if post is not None:
            Wc, bc, A, Bm = post
            b._f32('planar.Wc', Wc, (4 * (cin + h), h))
            if bc is not None:
                b._f32('planar.bc', bc, (h,))
            for name, t in (('A', A), ('Bm', Bm)):
                self._pl('planar.' + name, t, (R, Cc, h))
        b._same_device(X, H, SX, SH, Tc, W, bias, U, Rg, RH, Wc, bc, A, Bm)
        b._launch('stc_cell_gates_fwd_planar_bf16', H, _ptr(X), _ptr(H), _ptr(SX), _ptr(SH), _ptr(Tc), _ptr(W), _ptr(bias),
                  _ptr(U), _ptr(Rg), _ptr(RH), _ptr(Wc), _ptr(bc), _ptr(A), _ptr(Bm), R, Cc, cin + h, h,
                     nbytes=2 * R * Cc * (2 * cin + 2 * h + h * (2 + (RH is not None) + (2 if post is not None else 0))))

    def node_post_fwd(self, *a, **kw):
        raise StcError('bf16 planar path: the candidate projection runs inside cell_gates_fwd_planar (post=); STC_FUSE_POST=0 is an fp32-path switch')

    def spmm_blend_fwd(self, rowptr, colidx, val, plan, Bm, A, U, H, Cand, Hnew, copies=(), side=None):
        b = self.b
        if copies or side is not None:
            raise StcError('bf16 planar path: state copies into interleaved rows do not exist (every cell reads planes)')
        B, n, Cc, h = H.shape
        for name, t in (('Bm', Bm), ('A', A), ('U', U), ('H', H), ('Cand', Cand), ('Hnew', Hnew)):
            self._pl('spmm_blend.' + name, t, (B, n, Cc, h))
        b._same_device(rowptr, colidx, val, Bm, A, U, H, Cand, Hnew)
        g = b._graph_ptrs(rowptr, colidx, val, plan, n)
        b._launch('stc_spmm_blend_fwd_bf16', H, *g, n, n, _ptr(Bm), _ptr(A), _ptr(U), _ptr(H), _ptr(Cand), _ptr(Hnew), B, Cc, h,
                  nbytes=colidx.numel() * 8 + 4 * (n + 1) + 2 * B * n * Cc * h * 6)

    def spmm_sum(self, rowptr, colidx, val, plan, X, X2, addends, Y, blend=None):
        b = self.b
        B, n, Cc, h = Y.shape
        self._pl('spmm_sum.Y', Y, (B, n, Cc, h))
        self._pl('spmm_sum.X', X, (B, n, Cc, h))
        if X2 is not None:
            self._pl('spmm_sum.X2', X2, (B, n, Cc, h))
        if len(addends) > 5:
            raise StcError(f'spmm_sum: at most five addends, got {len(addends)}')
        ptrs = (_p * 5)()
        for i, (t, off) in enumerate(addends):
            self._pl(f'spmm_sum.add{i}', t, (B, n, Cc, h))
            if off != 0:
                raise StcError('spmm_sum (bf16): addends are whole planes')
            ptrs[i] = t.data_ptr()
        U = Cand = dY = None
        if blend is not None:
            U, Cand, dY = blend
            for name, t in (('U', U), ('Cand', Cand), ('dY', dY)):
                self._pl('spmm_sum.' + name, t, (B, n, Cc, h))
        b._same_device(rowptr, colidx, val, X, X2, Y, U, Cand, dY, *[t for t, _ in addends])
        g = b._graph_ptrs(rowptr, colidx, val, plan, n)
        b._launch('stc_spmm_sum_bf16', Y, *g, n, n, _ptr(X), _ptr(X2), len(addends), ptrs, _ptr(Y), _ptr(U), _ptr(Cand), _ptr(dY), B, Cc, h,
                  nbytes=colidx.numel() * 8 + 4 * (n + 1) + 2 * B * n * Cc * h * (2 + (X2 is not None) + len(addends) + (3 if blend else 0)))

    def gru_blend_bwd(self, dHnew, U, H, Cand, dCpre, dU, dH):
        if dU is not None or dH is not None:
            raise StcError('gru_blend_bwd (bf16): only the dCpre form exists')
        for name, t in (('dHnew', dHnew), ('U', U), ('Cand', Cand), ('dCpre', dCpre)):
            self._pl('blend_bwd.' + name, t, tuple(dHnew.shape))
        self.b._launch('stc_gru_blend_bwd_bf16', dHnew, _ptr(dHnew), _ptr(U), _ptr(Cand), _ptr(dCpre), dHnew.numel())

    def node_post_bwd(self, X, Tc, W, dA, dB, dX, dW, db, X2=None, dX2=None):
        b = self.b
        if X2 is None:
            raise StcError('node_post_bwd (bf16): planar form only (X2)')
        R, Cc, h = X.shape
        w2 = X2.shape[-1]
        narrow = w2 != 16
        if (dX2 is not None) == narrow:
            raise StcError('post: a planar gradient (dX2) goes with a 16 + 16 planar input and only with it')
        self._pl('post.X', X, (R, Cc, 16))
        self._pl('post.X2', X2, (R, Cc, w2))
        for name, t in (('dA', dA), ('dB', dB), ('dX', dX)) + ((('dX2', dX2),) if dX2 is not None else ()):
            self._pl('post.' + name, t, (R, Cc, 16))
        b._f32('post.Tc', Tc, (2, Cc, Cc))
        b._f32('post.W', W, (4 * (16 + w2), 16))
        b._f32('post.dW', dW, (4 * (16 + w2), 16))
        if db is not None:
            b._f32('post.db', db, (16,))
        b._same_device(X, X2, Tc, W, dA, dB, dX, dX2, dW, db)
        ws = b._get_workspace(X.device, b.lib.stc_bdg_node_bwd_workspace_bytes(2, 2, Cc, 32, 16, 0))
        b._launch('stc_bdg_node_post_bwd_bf16', X, _ptr(X), _ptr(X2), _ptr(Tc), _ptr(W), _ptr(dA), _ptr(dB), _ptr(dX), _ptr(dX2), _ptr(dW), _ptr(db),
                  _ptr(ws), ws.numel(), R, Cc, 16 + w2, 16,
                  nbytes=2 * R * Cc * (16 + w2 + 2 * 16 + 16 + (16 if dX2 is not None else 0)))

    def cell_bwd_planar_supported(self, Cc, h, cin=16) -> bool:
        """Whether the one-launch backward exists for cells with an input plane of ``cin`` columns (C = 64: the wide input only)."""
        return bool(self.b.lib.stc_cell_bwd_planar_bf16_supported(Cc, cin + h, h))

    def cell_bwd_planar(self, X, H, SX, SH, Tc, Wg, Wc, U, Rg, Cand, dHnew, dBm, dZs, dWg, dbg, dWc, dbc, accumulate_x=False, accumulate_h=False):
        """stc_cell_bwd_planar_bf16: candidate + gates backward of one planar cell step in one launch (no accumulate forms on bf16 planes)."""
        b = self.b
        if accumulate_x or accumulate_h:
            raise StcError('cell backward (bf16): the accumulate forms are fp32-path features')
        R, Cc, h, cin = self._planes(X, H, SX, SH)
        b._f32('cell_bwd.Tc', Tc, (2, Cc, Cc))
        b._f32('cell_bwd.Wg', Wg, (4 * (cin + h), 2 * h))
        b._f32('cell_bwd.Wc', Wc, (4 * (cin + h), h))
        for name, t in (('U', U), ('Rg', Rg), ('Cand', Cand), ('dHnew', dHnew), ('dBm', dBm)):
            self._pl('cell_bwd.' + name, t, (R, Cc, h))
        if len(dZs) != 4:
            raise StcError('cell backward: four gradient planes (dX, dSX, dH, dSH)')
        for i, z in enumerate(dZs):
            if z is None and i < 2 and cin != h:
                continue
            self._pl(f'cell_bwd.dZ[{i}]', z, (R, Cc, h))
        b._f32('cell_bwd.dWg', dWg, (4 * (cin + h), 2 * h))
        b._f32('cell_bwd.dWc', dWc, (4 * (cin + h), h))
        if dbg is not None:
            b._f32('cell_bwd.dbg', dbg, (2 * h,))
        if dbc is not None:
            b._f32('cell_bwd.dbc', dbc, (h,))
        b._same_device(X, H, SX, SH, Tc, Wg, Wc, U, Rg, Cand, dHnew, dBm, *dZs, dWg, dbg, dWc, dbc)
        ws = b._get_workspace(H.device, b.lib.stc_bdg_node_bwd_workspace_bytes(2, 2, Cc, 32, 32, 0) + b.lib.stc_bdg_node_bwd_workspace_bytes(2, 2, Cc, 32, 16, 0))
        b._launch('stc_cell_bwd_planar_bf16', H, _ptr(X), _ptr(H), _ptr(SX), _ptr(SH), _ptr(Tc), _ptr(Wg), _ptr(Wc), _ptr(U), _ptr(Rg), _ptr(Cand),
                  _ptr(dHnew), _ptr(dBm), *[_ptr(z) for z in dZs], _ptr(dWg), _ptr(dbg), _ptr(dWc), _ptr(dbc), _ptr(ws), ws.numel(), R, Cc, cin + h, h,
                  nbytes=2 * R * Cc * (2 * cin + 7 * h + 2 * h + (2 * h if cin == h else 0)), tag='wide' if cin == h else 'layer0')

    def cell_gates_bwd_planar(self, X, H, SX, SH, Tc, W, dRH, Cand, U, Rg, dHnew, dZs, dW, db, dH):
        b = self.b
        R, Cc, h, cin = self._planes(X, H, SX, SH)
        b._f32('planar.Tc', Tc, (2, Cc, Cc))
        b._f32('planar.W', W, (4 * (cin + h), 2 * h))
        for name, t in (('dRH', dRH), ('Cand', Cand), ('U', U), ('Rg', Rg), ('dHnew', dHnew)) + ((('dH', dH),) if dH is not None else ()):
            self._pl('planar.' + name, t, (R, Cc, h))                 # dH None: folded into dZs[2] by the kernel
        if len(dZs) != 4:
            raise StcError('planar gates backward: four gradient planes (dX, dSX, dH, dSH)')
        for i, z in enumerate(dZs):
            if z is None and i < 2 and cin != h:
                continue
            self._pl(f'planar.dZ[{i}]', z, (R, Cc, h))
        b._f32('planar.dW', dW, (4 * (cin + h), 2 * h))
        if db is not None:
            b._f32('planar.db', db, (2 * h,))
        b._same_device(X, H, SX, SH, Tc, W, dRH, Cand, U, Rg, dHnew, *dZs, dW, db, dH)
        ws = b._get_workspace(H.device, b.lib.stc_bdg_node_bwd_workspace_bytes(2, 2, Cc, 2 * h, 2 * h, 0))
        zp = (_p * 4)(*[0 if z is None else z.data_ptr() for z in dZs])
        b._launch('stc_cell_gates_bwd_planar_bf16', H, _ptr(X), _ptr(H), _ptr(SX), _ptr(SH), _ptr(Tc), _ptr(W), _ptr(dRH), _ptr(Cand),
                  _ptr(U), _ptr(Rg), _ptr(dHnew), zp, _ptr(dW), _ptr(db), _ptr(dH), _ptr(ws), ws.numel(), R, Cc, cin + h, h,
                  nbytes=2 * R * Cc * (2 * cin + 2 * h + 5 * h + 2 * h + (2 * h if cin == h else 0) + (h if dH is not None else 0)))
