"""Host side of the MI355X-native STC-GNN message-passing path.

``stc-gnn_amd/`` is the directory to put on ``sys.path`` in place of the
reference's ``framework/``: it provides the top-level module ``STC_GNN`` (same
classes, constructors, ``forward`` signatures and ``state_dict`` keys as
``framework/STC_GNN.py``) whose arithmetic runs in hand-written HIP kernels
(``csrc/`` -> ``libstc_hip.so``, C ABI in ``include/stc_hip.h``).

    stc_hip._lib    ctypes binding of the C ABI, host-side argument checks
    stc_hip.graph   CSR containers for the spatial graph (fixed sparse / learned dense)
    stc_hip.ops     autograd operators = host sequences of kernel launches
    stc_hip.dist    batch-sharded training: RCCL all-reduce of the gradient bucket
"""
from .graph import CsrGraph  # noqa: F401
from ._lib import StcError, LIB_PATH  # noqa: F401

__all__ = ['CsrGraph', 'StcError', 'LIB_PATH']
