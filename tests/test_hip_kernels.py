"""-m gpu: every exported HIP kernel against its CPU twin (oracle/kernel_emul.py), called through
the C ABI (ctypes front ``stc_hip._lib.HipKernels``) on the same seeded inputs.

Tolerance: 1e-5 relative (max|a-b| / max|b|), the bound BASELINE.json's north_star states for
fp32; observed differences are summation-order noise around 1e-6.  Index outputs do not exist on
this path (all results are fp32).  Edge cases: empty rows, rows longer than the LDS-staged
segment, ragged last tiles, widths that are not multiples of 4 (SF shape F = 85), zero sizes,
in-place epilogue.
"""
import numpy as np
import pytest
import torch

from oracle.kernel_emul import EmulatedKernels
from stc_hip._lib import KernelTimer
from stc_hip import CsrGraph
from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5
EM = EmulatedKernels()


@pytest.fixture(scope='module')
def hip():
    from stc_hip._lib import HipKernels
    return HipKernels()


def cu(t):
    return None if t is None else t.cuda()


def random_csr(n_rows, n_cols, density, seed, empty_rows=()):
    g = torch.Generator().manual_seed(seed)
    mask = torch.rand(n_rows, n_cols, generator=g) < density
    for r in empty_rows:
        mask[r] = False
    vals = torch.randn(n_rows, n_cols, generator=g) * mask
    idx = mask.nonzero(as_tuple=False)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(mask.sum(1), 0)
    return rowptr.to(torch.int32), idx[:, 1].to(torch.int32).contiguous(), vals[mask].contiguous(), vals


@pytest.mark.parametrize('n_rows,n_cols,F,B,density', [
    (37, 37, 1024, 2, 0.2),      # vector path VPT=4, ragged last tile
    (64, 50, 160, 3, 0.3),       # vector path VPT=1 (SF layer>=1: C*L = 5*32)
    (100, 100, 85, 2, 1.0),      # scalar path, dense rows (SF layer 0: C*L = 5*17)
    (40, 40, 512, 1, 1.0),       # vector path VPT=2
    (300, 300, 256, 1, 1.0),     # rows longer than the staged segment (8 rows x 300 > 1024 entries)
    (9, 9, 2048, 1, 0.5),        # two column blocks per row
    (5, 7, 3, 1, 0.5),           # tiny scalar
])
@pytest.mark.parametrize('alpha,beta', [(1.0, 0.0), (2.0, -1.0)])
def test_csr_spmm(hip, n_rows, n_cols, F, B, density, alpha, beta):
    rowptr, colidx, val, dense = random_csr(n_rows, n_cols, density, seed=n_rows * 7 + F, empty_rows=(1, n_rows - 1))
    g = torch.Generator().manual_seed(F)
    X = torch.randn(B, n_cols, F, generator=g)
    Y0 = torch.randn(B, n_rows, F, generator=g) if beta != 0 else None
    want = torch.empty(B, n_rows, F)
    EM.csr_spmm(rowptr, colidx, val, n_rows, n_cols, X, Y0, want, alpha, beta)
    dense_want = alpha * torch.einsum('rc,bcf->brf', dense, X) + (beta * Y0 if Y0 is not None else 0)
    assert rel_err(want, dense_want) < 1e-5                      # the twin itself vs a dense matmul
    got = torch.full((B, n_rows, F), float('nan')).cuda()
    hip.csr_spmm(cu(rowptr), cu(colidx), cu(val), n_rows, n_cols, cu(X), cu(Y0), got, alpha, beta)
    assert rel_err(got, want) < TOL
    if Y0 is not None:                                          # in-place epilogue: Y0 aliases Y
        buf = Y0.clone().cuda()
        hip.csr_spmm(cu(rowptr), cu(colidx), cu(val), n_rows, n_cols, cu(X), buf, buf, alpha, beta)
        assert torch.equal(buf, got)


def _banded_graph(n, half_width, seed):
    g = torch.Generator().manual_seed(seed)
    i = torch.arange(n)
    M = ((i[:, None] - i[None, :]).abs() <= half_width) & (torch.rand(n, n, generator=g) < 0.7)
    M[3] = False                                                  # an empty row
    V = torch.randn(n, n, generator=g) * M
    return CsrGraph.from_dense(V), V


@pytest.mark.parametrize('n,F,B,hw', [(203, 1024, 1, 4), (77, 640, 2, 3), (64, 256, 1, 6), (1001, 64, 1, 2), (30, 2048, 1, 40),
                                      (1001, 32, 3, 2), (203, 16, 1, 4), (77, 8, 2, 3), (50, 4, 1, 40), (4099, 32, 2, 1)])
def test_bcsr_spmm_equals_csr(hip, n, F, B, hw):
    """Row-blocked kernel (4 output rows per wave, distinct neighbour rows fetched once per block) vs the CSR kernel
    vs a dense matmul; ragged last block (n % 4 != 0), an empty row, block lists longer than the LDS-staged
    segment (dense band, hw=40), two column blocks (F=2048), in-place beta epilogue, both graph orientations; narrow rows of
    4 .. 32 floats (the layer-0 input plane: several row blocks per wave, spmm_bcsr_narrow_kernel)."""
    graph, V = _banded_graph(n, hw, seed=n + F)
    d = graph.on(torch.device('cuda'))
    g = torch.Generator().manual_seed(F)
    X = torch.randn(B, n, F, generator=g)
    Y0 = torch.randn(B, n, F, generator=g)
    for side, dense in (('fwd', V.t()), ('bwd', V)):
        rp, ci, vals = d[f'{side}_rowptr'], d[f'{side}_colidx'], d[f'{side}_val']
        plan = (d[f'{side}_blk_ptr'], d[f'{side}_blk_cols'], d[f'{side}_blk_vals'])
        for alpha, beta in ((1.0, 0.0), (2.0, -1.0)):
            want = alpha * torch.einsum('rc,bcf->brf', dense, X) + beta * Y0
            csr = Y0.clone().cuda()
            hip.csr_spmm(rp, ci, vals, n, n, cu(X), csr if beta else None, csr, alpha, beta)
            blocked = Y0.clone().cuda()
            hip.csr_spmm(rp, ci, vals, n, n, cu(X), blocked if beta else None, blocked, alpha, beta, plan=plan)
            assert rel_err(csr, want) < TOL and rel_err(blocked, want) < TOL
            assert rel_err(blocked, csr) < 2e-6


@pytest.mark.parametrize('n,F,B', [(100, 160, 32), (100, 100, 3), (12, 20, 2), (200, 256, 4), (37, 7, 1), (300, 64, 2), (16, 16, 1)])
@pytest.mark.parametrize('alpha,beta', [(1.0, 0.0), (2.0, -1.0)])
def test_dense_graph_aggregation(hip, n, F, B, alpha, beta):
    """stc_dense_agg_f32 (exact-fp32 matrix cores): Y = alpha S.X + beta Y0 with a dense S -- what ``csr_spmm`` launches when it is handed the
    full N x N pattern of a learned graph -- against the dense product in float64 and against the CSR kernel on the same values; sizes
    off every tile boundary (rows / graph columns not multiples of 16 / 4, feature columns not a multiple of 16), in place (Y0 = Y)."""
    from stc_hip.graph import full_pattern, is_full_pattern
    g = torch.Generator().manual_seed(n + F)
    S = torch.softmax(torch.randn(n, n, generator=g), -1)
    X, Y0 = torch.randn(B, n, F, generator=g), torch.randn(B, n, F, generator=g)
    want = (alpha * torch.einsum('rc,bcf->brf', S.double(), X.double()) + beta * Y0.double()).float()
    rowptr, colidx = full_pattern(n, torch.device('cuda'))
    assert is_full_pattern(colidx, n, n) and not is_full_pattern(colidx.clone(), n, n)
    val = cu(S).reshape(-1)
    Y = Y0.clone().cuda()
    hip.timer = KernelTimer()
    try:
        hip.csr_spmm(rowptr, colidx, val, n, n, cu(X), Y if beta else None, Y, alpha, beta)          # in place when there is a Y0
        names = set(hip.timer.summary())
    finally:
        hip.timer = None
    assert names == {'stc_dense_agg_f32'}, names
    assert rel_err(Y, want) < TOL
    Yc = Y0.clone().cuda()
    hip.csr_spmm(rowptr, colidx.clone(), val, n, n, cu(X), Yc if beta else None, Yc, alpha, beta)     # same values through the CSR kernel
    assert rel_err(Y, Yc) < 2e-6


def test_row_block_plan_fetch_counts():
    assert CsrGraph.queen_grid(40, 40).fetches_per_row[0] < 5.0                      # 18 fetches per 4 rows in the interior
    assert CsrGraph.queen_grid(40, 40, permute_seed=1).fetches_per_row[0] > 7.0     # random node order: nothing to share


def test_csr_spmm_zero_sizes_and_errors(hip):
    from stc_hip._lib import StcError
    rowptr = torch.zeros(1, dtype=torch.int32).cuda()
    empty_i = torch.zeros(0, dtype=torch.int32).cuda()
    empty_f = torch.zeros(0).cuda()
    hip.csr_spmm(rowptr, empty_i, empty_f, 0, 4, torch.randn(2, 4, 8).cuda(), None, torch.empty(2, 0, 8).cuda(), 1.0, 0.0)
    rp = torch.zeros(5, dtype=torch.int32).cuda()               # 4 empty rows -> zeros
    Y = torch.full((1, 4, 64), 7.0).cuda()
    hip.csr_spmm(rp, empty_i, empty_f, 4, 4, torch.randn(1, 4, 64).cuda(), None, Y, 1.0, 0.0)
    assert float(Y.abs().max()) == 0.0
    with pytest.raises(StcError):                               # CPU tensor: no fallback
        hip.csr_spmm(rp.cpu(), empty_i, empty_f, 4, 4, torch.randn(1, 4, 64), None, torch.empty(1, 4, 64), 1.0, 0.0)
    with pytest.raises(StcError):                               # wrong dtype
        hip.csr_spmm(rp, empty_i, empty_f, 4, 4, torch.randn(1, 4, 64).double().cuda(), None, Y, 1.0, 0.0)
    with pytest.raises(StcError):                               # shape mismatch caught on the host
        hip.csr_spmm(rp, empty_i, empty_f, 4, 4, torch.randn(1, 5, 64).cuda(), None, Y, 1.0, 0.0)
    with pytest.raises(StcError):                               # beta without Y0: rejected by the C side
        hip.csr_spmm(rp, empty_i, empty_f, 4, 4, torch.randn(1, 4, 64).cuda(), None, Y, 1.0, 1.0)


@pytest.mark.parametrize('n,F,B,density', [(33, 85, 2, 1.0), (50, 256, 3, 0.2), (12, 7, 1, 0.6)])
def test_csr_sddmm(hip, n, F, B, density):
    rowptr, colidx, val, _ = random_csr(n, n, density, seed=n + F, empty_rows=(0,))
    g = torch.Generator().manual_seed(n)
    A, Bm = torch.randn(B, n, F, generator=g), torch.randn(B, n, F, generator=g)
    base = torch.randn(colidx.numel(), generator=g)
    for acc in (False, True):
        want = base.clone()
        EM.csr_sddmm(rowptr, colidx, n, n, A, Bm, want, 2.0, acc)
        got = base.clone().cuda()
        hip.csr_sddmm(cu(rowptr), cu(colidx), n, n, cu(A), cu(Bm), got, 2.0, acc)
        assert rel_err(got, want) < TOL


@pytest.mark.parametrize('n,K', [(3, 1), (5, 2), (8, 3), (32, 4), (64, 3)])
def test_cheby_dense_fwd_bwd(hip, n, K):
    g = torch.Generator().manual_seed(n * 10 + K)
    G = torch.randn(n, n, generator=g) / n ** 0.5
    T_want = torch.empty(K, n, n)
    EM.cheby_dense_fwd(G, K, T_want)
    T = torch.empty(K, n, n).cuda()
    hip.cheby_dense_fwd(cu(G), K, T)
    assert rel_err(T, T_want) < TOL
    dT = torch.randn(K, n, n, generator=g)
    dG_want = torch.empty(n, n)
    EM.cheby_dense_bwd(G, T_want, dT.clone(), dG_want)
    dG = torch.empty(n, n).cuda()
    hip.cheby_dense_bwd(cu(G), T, cu(dT.clone()), dG)
    assert rel_err(dG, dG_want) < TOL


NODE_SHAPES = [
    # nodes, C, L, Ho, Ks, Kc
    (24, 3, 5, 4, 1, 1),
    (24, 3, 5, 4, 2, 2),
    (24, 3, 5, 4, 3, 3),
    (200, 5, 17, 32, 2, 2),      # SF encoder layer 0 gates
    (200, 5, 32, 16, 2, 2),      # SF candidate
    (70, 8, 32, 32, 3, 3),       # config 2 width, K=3, ragged tile (TN=4)
    (33, 32, 32, 32, 2, 2),      # headline width C=32
    (17, 32, 17, 16, 3, 2),      # Ks != Kc
    (5, 64, 32, 32, 2, 2),       # config 5 category count
    (9, 7, 6, 3, 4, 4),          # highest supported order
    # MFMA fast path: C in {16,32,64}, Ho in {16,32}, L in {20,32}, Ks = Kc in {1,2,3}
    (50, 32, 32, 16, 2, 2),      # candidate conv at the headline width
    (50, 32, 20, 32, 2, 2),      # encoder layer 0 (in=1 padded to L=20), gates
    (50, 32, 20, 16, 2, 2),      # encoder layer 0, candidate
    (21, 16, 32, 32, 2, 2),
    (21, 16, 20, 16, 3, 3),
    (13, 64, 20, 16, 2, 2),
    (13, 32, 32, 32, 3, 3),      # config 4 order
    (13, 32, 32, 16, 1, 1),
    (13, 32, 20, 32, 1, 1),      # split-operand path corner shapes: one slab, odd (c, o) block counts, C = 64
    (21, 32, 20, 16, 3, 3),
    (7, 64, 32, 32, 1, 1),
    (9, 64, 32, 16, 2, 2),
    (4500, 32, 32, 32, 2, 2),    # more nodes than resident waves: grid-stride + cross-workgroup reduction
    (4500, 32, 20, 16, 2, 2),
]


def _node_inputs(nodes, C, L, Ho, Ks, Kc, seed):
    g = torch.Generator().manual_seed(seed)
    Zs = [torch.randn(nodes, C, L, generator=g) for _ in range(Ks)]
    Tc = torch.randn(Kc, C, C, generator=g) / C ** 0.5
    Tc[0] = torch.eye(C)
    W = torch.randn(Ks * Kc * L, Ho, generator=g) / (Ks * Kc * L) ** 0.5
    b = torch.randn(Ho, generator=g)
    dY = torch.randn(nodes, C, Ho, generator=g)
    return Zs, Tc, W, b, dY


@pytest.fixture(params=['default', 'fp32-mfma', 'generic-only'])
def node_path(request, hip):
    """Run every node-kernel case three times: default dispatch (split-operand MFMA where the shape allows, then fp32 MFMA, then
    generic), fp32 MFMA or generic only, generic VALU only (stc_set_dispatch_level: the library reads no environment)."""
    hip.set_dispatch_level({'default': 0, 'fp32-mfma': 1, 'generic-only': 2}[request.param])
    yield request.param
    hip.set_dispatch_level(0)


@pytest.fixture(params=['default', 'fp32-mfma'])
def fused_path(request, hip):
    """The fused cell kernels exist in both matrix-core flavours."""
    hip.set_dispatch_level({'default': 0, 'fp32-mfma': 1}[request.param])
    yield request.param
    hip.set_dispatch_level(0)


@pytest.mark.parametrize('shape', NODE_SHAPES)
@pytest.mark.parametrize('bias', [True, False])
def test_bdg_node_fwd(hip, shape, bias, node_path):
    nodes, C, L, Ho, Ks, Kc = shape
    Zs, Tc, W, b, _ = _node_inputs(*shape, seed=sum(shape))
    b = b if bias else None
    want = torch.empty(nodes, C, Ho)
    EM.bdg_node_fwd(Zs, Tc, W, b, want)
    got = torch.full((nodes, C, Ho), float('nan')).cuda()
    hip.bdg_node_fwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(b), got)
    assert rel_err(got, want) < TOL


@pytest.mark.parametrize('shape', NODE_SHAPES)
@pytest.mark.parametrize('want_dT', [True, False])
def test_bdg_node_bwd(hip, shape, want_dT, node_path):
    nodes, C, L, Ho, Ks, Kc = shape
    Zs, Tc, W, b, dY = _node_inputs(*shape, seed=sum(shape) + 1)
    dZ_w = [torch.empty(nodes, C, L) for _ in range(Ks)]
    dW_w, db_w = torch.empty_like(W), torch.empty(Ho)
    dT_w = torch.empty_like(Tc) if want_dT else None
    EM.bdg_node_bwd(Zs, Tc, W, dY, dZ_w, dW_w, db_w, dT_w)
    nan = float('nan')
    dZ = [torch.full((nodes, C, L), nan).cuda() for _ in range(Ks)]
    dW, db = torch.full_like(W, nan).cuda(), torch.full((Ho,), nan).cuda()
    dT = torch.full_like(Tc, nan).cuda() if want_dT else None
    hip.bdg_node_bwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(dY), dZ, dW, db, dT)
    for a, w in zip(dZ, dZ_w):
        assert rel_err(a, w) < TOL
    assert rel_err(dW, dW_w) < TOL and rel_err(db, db_w) < TOL
    if want_dT:
        assert float(dT[0].abs().max()) == 0.0                  # T_0 = I is a constant
        if Kc > 1:
            assert rel_err(dT[1:], dT_w[1:]) < TOL
    # bitwise reproducible: partial sums are combined in a fixed order
    dW2 = torch.empty_like(dW)
    hip.bdg_node_bwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(dY), dZ, dW2, db, dT)
    assert torch.equal(dW, dW2)


@pytest.mark.parametrize('nodes,C,L,Lw,Ho,K', [
    (50, 32, 20, 17, 32, 2),     # encoder layer 0 at the headline width: in + hidden = 17, rows padded to 20 (MFMA path)
    (50, 32, 20, 17, 16, 2),
    (4100, 32, 20, 17, 32, 2),
    (21, 16, 20, 18, 16, 3),
    (40, 5, 20, 17, 32, 2),      # SF shape, padded (generic path)
    (30, 3, 8, 5, 4, 3),
])
def test_bdg_node_padded_feature_rows(hip, nodes, C, L, Lw, Ho, K, node_path):
    """Slabs carry L - Lw zero-pad columns; W has Lw rows per block; pad columns get zero gradient."""
    g = torch.Generator().manual_seed(nodes + L + Lw)
    Zs = [torch.randn(nodes, C, L, generator=g) for _ in range(K)]
    for z in Zs:
        z[..., Lw:] = 7.0                                         # garbage in the pad columns must not matter
    Tc = torch.randn(K, C, C, generator=g) / C ** 0.5
    Tc[0] = torch.eye(C)
    W = torch.randn(K * K * Lw, Ho, generator=g) / (K * K * Lw) ** 0.5
    b = torch.randn(Ho, generator=g)
    dY = torch.randn(nodes, C, Ho, generator=g)
    Y_w = torch.empty(nodes, C, Ho)
    EM.bdg_node_fwd(Zs, Tc, W, b, Y_w)
    Y = torch.empty(nodes, C, Ho).cuda()
    hip.bdg_node_fwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(b), Y)
    assert rel_err(Y, Y_w) < TOL
    dZ_w = [torch.empty(nodes, C, L) for _ in range(K)]
    dW_w, db_w = torch.empty_like(W), torch.empty(Ho)
    EM.bdg_node_bwd(Zs, Tc, W, dY, dZ_w, dW_w, db_w, None)
    dZ = [torch.full((nodes, C, L), float('nan')).cuda() for _ in range(K)]
    dW, db = torch.empty_like(W).cuda(), torch.empty(Ho).cuda()
    hip.bdg_node_bwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(dY), dZ, dW, db, None)
    for a, w in zip(dZ, dZ_w):
        assert rel_err(a, w) < TOL
        assert float(a[..., Lw:].abs().max()) == 0.0
    assert rel_err(dW, dW_w) < TOL and rel_err(db, db_w) < TOL


@pytest.mark.parametrize('nodes,C,cin,K', [(50, 32, 16, 2), (50, 32, 1, 2), (21, 16, 16, 3), (13, 64, 1, 2), (4500, 32, 16, 2),
                                            (4100, 32, 1, 2), (9, 32, 16, 1)])
def test_fused_cell_epilogues(hip, nodes, C, cin, K, fused_path):
    """Gate math fused into the node kernel's epilogue (hidden 16) vs node kernel + gate kernels of the CPU twin."""
    h = 16
    Lw = cin + h
    L = Lw + (-Lw) % 4
    assert hip.cell_fused_supported(K, K, C, L, h)
    g = torch.Generator().manual_seed(nodes + C + cin + K)
    Zs = [torch.randn(nodes, C, L, generator=g) for _ in range(K)]
    Zs[0][..., Lw:] = 0.0                                         # pad columns of the concatenated input are zeros
    Tc = torch.randn(K, C, C, generator=g) / C ** 0.5
    Tc[0] = torch.eye(C)
    H = torch.randn(nodes, C, h, generator=g)
    Wg = torch.randn(K * K * Lw, 2 * h, generator=g) / (K * K * Lw) ** 0.5
    bg = torch.randn(2 * h, generator=g)
    U_w, R_w, Ci_w = torch.empty_like(H), torch.empty_like(H), torch.empty(nodes, C, L)
    EM.cell_gates_fwd(Zs, Tc, Wg, bg, H, U_w, R_w, Ci_w)
    U, R, Ci = (torch.full(t.shape, float('nan')).cuda() for t in (U_w, R_w, Ci_w))
    hip.cell_gates_fwd([cu(z) for z in Zs], cu(Tc), cu(Wg), cu(bg), cu(H), U, R, Ci)
    assert rel_err(U, U_w) < TOL and rel_err(R, R_w) < TOL and rel_err(Ci, Ci_w) < TOL
    assert float(Ci[..., Lw:].abs().max()) == 0.0 if L > Lw else True
    assert torch.equal(Ci[..., :cin].cpu(), Zs[0][..., :cin])    # Xt is copied bit for bit

    Wc = torch.randn(K * K * Lw, h, generator=g) / (K * K * Lw) ** 0.5
    bc = torch.randn(h, generator=g)
    Cand_w, Hn_w = torch.empty_like(H), torch.empty_like(H)
    EM.cell_blend_fwd(Zs, Tc, Wc, None, U_w, H, Cand_w, Hn_w)
    Cand, Hn = torch.full(H.shape, float('nan')).cuda(), torch.full(H.shape, float('nan')).cuda()
    hip.cell_blend_fwd([cu(z) for z in Zs], cu(Tc), cu(Wc), None, cu(U_w), cu(H), Cand, Hn)
    assert rel_err(Cand, Cand_w) < TOL and rel_err(Hn, Hn_w) < TOL
    EM.cell_blend_fwd(Zs, Tc, Wc, bc, U_w, H, Cand_w, Hn_w)
    hip.cell_blend_fwd([cu(z) for z in Zs], cu(Tc), cu(Wc), cu(bc), cu(U_w), cu(H), Cand, Hn)
    assert rel_err(Cand, Cand_w) < TOL and rel_err(Hn, Hn_w) < TOL


@pytest.mark.parametrize('nodes,C,L,Lw', [(50, 32, 32, 32), (50, 32, 20, 17), (13, 64, 32, 32), (9, 64, 20, 20), (4500, 32, 32, 32)])
def test_post_aggregation_backward(hip, nodes, C, L, Lw):
    """stc_bdg_node_post_bwd_f32: backward of Y = A + S.Bm from (X, dA, dBm) vs the CPU twin; bitwise reproducible dW."""
    Ho, K = 16, 2
    assert hip.node_post_supported(K, K, C, L, Ho)
    g = torch.Generator().manual_seed(nodes + C + L)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    X = rnd(nodes, C, L)
    X[..., Lw:] = 7.0                                               # garbage in the pad columns must not matter
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    W = rnd(K * K * Lw, Ho) / (K * K * Lw) ** 0.5
    dA, dB = rnd(nodes, C, Ho), rnd(nodes, C, Ho)
    dX_w, dW_w, db_w = torch.empty(nodes, C, L), torch.empty_like(W), torch.empty(Ho)
    EM.node_post_bwd(X, Tc, W, dA, dB, dX_w, dW_w, db_w)
    nan = float('nan')
    dX, dW, db = torch.full((nodes, C, L), nan).cuda(), torch.full_like(W, nan).cuda(), torch.full((Ho,), nan).cuda()
    hip.node_post_bwd(cu(X), cu(Tc), cu(W), cu(dA), cu(dB), dX, dW, db)
    assert rel_err(dX, dX_w) < TOL and rel_err(dW, dW_w) < TOL and rel_err(db, db_w) < TOL
    assert float(dX[..., Lw:].abs().max()) == 0.0 if L > Lw else True
    dW2 = torch.empty_like(dW)
    hip.node_post_bwd(cu(X), cu(Tc), cu(W), cu(dA), cu(dB), dX, dW2, None)
    assert torch.equal(dW, dW2)


@pytest.mark.parametrize('batch,grid,C,L,Lw,copy_case', [(2, (5, 5), 32, 32, 32, 'pair'), (3, (4, 7), 32, 20, 17, 'side'), (1, (3, 3), 64, 32, 32, 'none'),
                                                        (2, (40, 56), 32, 32, 32, 'pair'), (1, (1, 1), 32, 20, 20, 'none')])
def test_post_aggregation_forward(hip, batch, grid, C, L, Lw, copy_case):
    """Candidate convolution as Y = A + S.Bm: stc_bdg_node_post_fwd_f32 then stc_spmm_blend_fwd_f32 (blend + state copies in the
    SpMM's epilogue) vs the CPU twin, and vs the slab form (SpMM first, fused node kernel) of the same convolution."""
    h, K = 16, 2
    graph = CsrGraph.queen_grid(*grid, normalize=True)
    n = graph.n
    nodes = batch * n
    g = torch.Generator().manual_seed(nodes + C + L)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    X = rnd(nodes, C, L)
    X[..., Lw:] = 0.0
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    W, b = rnd(K * K * Lw, h) / (K * K * Lw) ** 0.5, rnd(h)
    U, H = torch.rand(batch, n, C, h, generator=g), rnd(batch, n, C, h)
    A_w, B_w = torch.empty(nodes, C, h), torch.empty(nodes, C, h)
    EM.node_post_fwd(X, Tc, W, b, A_w, B_w)
    A, Bm = torch.full_like(A_w, float('nan')).cuda(), torch.full_like(B_w, float('nan')).cuda()
    hip.node_post_fwd(cu(X), cu(Tc), cu(W), cu(b), A, Bm)
    assert rel_err(A, A_w) < TOL and rel_err(Bm, B_w) < TOL

    hst = graph._host
    csr = tuple(torch.from_numpy(hst[k]) for k in ('fwd_rowptr', 'fwd_colidx', 'fwd_val'))
    dev = graph.on(torch.device('cuda'))
    plan = (dev['fwd_blk_ptr'], dev['fwd_blk_cols'], dev['fwd_blk_vals'])
    def bufs(where):
        mk = (lambda *s_: torch.full(s_, 9.0)) if where == 'cpu' else (lambda *s_: torch.full(s_, 9.0).cuda())
        if copy_case == 'pair':
            return [(mk(nodes, C, 32), 16), (mk(nodes, C, 32), 0)], None
        if copy_case == 'side':
            side = rnd(nodes, C, 1) if where == 'cpu' else None
            return [(mk(nodes, C, 20), 1)], side
        return [], None
    cp_w, side_w = bufs('cpu')
    Cand_w, Hn_w = torch.empty_like(H), torch.empty_like(H)
    EM.spmm_blend_fwd(*csr, None, B_w.view(batch, n, C, h), A_w.view(batch, n, C, h), U, H, Cand_w, Hn_w, copies=cp_w, side=side_w)
    cp, _ = bufs('cuda')
    Cand, Hn = torch.full_like(H, float('nan')).cuda(), torch.full_like(H, float('nan')).cuda()
    hip.spmm_blend_fwd(*(cu(t) for t in csr), plan, cu(B_w).view(batch, n, C, h), cu(A_w).view(batch, n, C, h), cu(U), cu(H), Cand, Hn,
                       copies=cp, side=None if side_w is None else cu(side_w))
    assert rel_err(Cand, Cand_w) < TOL and rel_err(Hn, Hn_w) < TOL
    for (got, _), (want, _) in zip(cp, cp_w):
        assert torch.equal(got.cpu() == 9.0, want == 9.0) and rel_err(got, want) < TOL      # untouched columns stay untouched
    # the same convolution in its slab form
    Z1 = torch.empty_like(X)
    EM.csr_spmm(*csr, n, n, X.view(batch, n, C * L), None, Z1.view(batch, n, C * L), 1.0, 0.0)
    Cand_s, Hn_s = torch.empty_like(H), torch.empty_like(H)
    EM.cell_blend_fwd([X, Z1], Tc, W, b, U.view(nodes, C, h), H.view(nodes, C, h), Cand_s.view(nodes, C, h), Hn_s.view(nodes, C, h))
    assert rel_err(Hn, Hn_s) < TOL and rel_err(Cand, Cand_s) < TOL


@pytest.mark.parametrize('nodes,C', [(50, 32), (13, 64), (4500, 32)])
def test_planar_cell_kernels(hip, nodes, C):
    """Planar cell inputs ([Xt | H] as two (nodes, C, 16) planes): gates forward / backward and the post-aggregation
    candidate kernels with a planar X, against the CPU twin (which concatenates the planes)."""
    h, K = 16, 2
    assert hip.cell_planar_supported(K, K, C, h)
    g = torch.Generator().manual_seed(nodes + C)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    X, H, SX, SH = (rnd(nodes, C, h) for _ in range(4))
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    Wg, bg = rnd(K * K * 2 * h, 2 * h) / (8 * h) ** 0.5, rnd(2 * h)
    U_w, R_w, RH_w = (torch.empty(nodes, C, h) for _ in range(3))
    EM.cell_gates_fwd_planar(X, H, SX, SH, Tc, Wg, bg, U_w, R_w, RH_w)
    nan = lambda *s_: torch.full(s_, float('nan')).cuda()
    U, R, RH = nan(nodes, C, h), nan(nodes, C, h), nan(nodes, C, h)
    hip.cell_gates_fwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(bg), U, R, RH)
    assert rel_err(U, U_w) < TOL and rel_err(R, R_w) < TOL and rel_err(RH, RH_w) < TOL

    dRH, Cand, dHn = rnd(nodes, C, h), torch.tanh(rnd(nodes, C, h)), rnd(nodes, C, h)
    dZ_w = [torch.empty(nodes, C, h) for _ in range(4)]                       # d X plane, d SX plane, d H plane, d SH plane
    dW_w, db_w, dH_w = torch.empty_like(Wg), torch.empty(2 * h), torch.empty(nodes, C, h)
    EM.cell_gates_bwd_planar(X, H, SX, SH, Tc, Wg, dRH, Cand, U_w, R_w, dHn, dZ_w, dW_w, db_w, dH_w)
    dZ = [nan(nodes, C, h) for _ in range(4)]
    dW, db, dH = nan(*Wg.shape), nan(2 * h), nan(nodes, C, h)
    hip.cell_gates_bwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(dRH), cu(Cand), cu(U_w), cu(R_w), cu(dHn), dZ, dW, db, dH)
    for a, w in zip(dZ, dZ_w):
        assert rel_err(a, w) < TOL
    assert rel_err(dW, dW_w) < TOL and rel_err(db, db_w) < TOL and rel_err(dH, dH_w) < TOL
    # dH = None: the kernel folds the state's share from the gate prologue into the H plane's gradient (lane-private LDS slot)
    dZf = [nan(nodes, C, h) for _ in range(4)]
    hip.cell_gates_bwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(dRH), cu(Cand), cu(U_w), cu(R_w), cu(dHn), dZf, dW, db, None)
    assert rel_err(dZf[2], dZ_w[2] + dH_w) < TOL and rel_err(dW, dW_w) < TOL
    for i in (0, 1, 3):
        assert torch.equal(dZf[i], dZ[i])

    Wc, bc = rnd(K * K * 2 * h, h) / (8 * h) ** 0.5, rnd(h)
    A_w, B_w = torch.empty(nodes, C, h), torch.empty(nodes, C, h)
    EM.node_post_fwd(X, Tc, Wc, bc, A_w, B_w, X2=RH_w)
    A, Bm = nan(nodes, C, h), nan(nodes, C, h)
    hip.node_post_fwd(cu(X), cu(Tc), cu(Wc), cu(bc), A, Bm, X2=cu(RH_w))
    assert rel_err(A, A_w) < TOL and rel_err(Bm, B_w) < TOL
    if hip.cell_planar_post_fused(C):          # the same pair from the gates launch itself (candidate projection as its second stage)
        U2, R2, RH2, A2, B2 = (nan(nodes, C, h) for _ in range(5))
        hip.cell_gates_fwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(bg), U2, R2, RH2, post=(cu(Wc), cu(bc), A2, B2))
        assert torch.equal(U2, U) and torch.equal(RH2, RH) and rel_err(A2, A_w) < TOL and rel_err(B2, B_w) < TOL
        hip.cell_gates_fwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(bg), U2, R2, RH2, post=(cu(Wc), None, A2, B2))
        assert rel_err(A2, A_w - bc) < TOL
    dA, dB = rnd(nodes, C, h), rnd(nodes, C, h)
    dX_w, dX2_w, dWc_w, dbc_w = torch.empty(nodes, C, h), torch.empty(nodes, C, h), torch.empty_like(Wc), torch.empty(h)
    EM.node_post_bwd(X, Tc, Wc, dA, dB, dX_w, dWc_w, dbc_w, X2=RH_w, dX2=dX2_w)
    dX, dX2, dWc, dbc = nan(nodes, C, h), nan(nodes, C, h), nan(*Wc.shape), nan(h)
    hip.node_post_bwd(cu(X), cu(Tc), cu(Wc), cu(dA), cu(dB), dX, dWc, dbc, X2=cu(RH_w), dX2=dX2)
    assert rel_err(dX, dX_w) < TOL and rel_err(dX2, dX2_w) < TOL and rel_err(dWc, dWc_w) < TOL and rel_err(dbc, dbc_w) < TOL


@pytest.mark.parametrize('nodes,C,cin', [(50, 32, 1), (13, 64, 4), (4500, 32, 1), (9, 32, 3)])
def test_planar_cell_kernels_narrow_input(hip, nodes, C, cin):
    """Layer-0 form of the planar kernels: a (nodes, C, cin) input plane with cin in 1..4 beside the 16-wide state plane; the
    kernels read the slab as [H | Xt | pad] with W's rows permuted inside -- results must equal the reference order."""
    h, K = 16, 2
    Lw = cin + h
    g = torch.Generator().manual_seed(nodes + C + cin)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    X, SX, H, SH = rnd(nodes, C, cin), rnd(nodes, C, cin), rnd(nodes, C, h), rnd(nodes, C, h)
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    Wg, bg = rnd(K * K * Lw, 2 * h) / (4 * Lw) ** 0.5, rnd(2 * h)
    U_w, R_w, RH_w = (torch.empty(nodes, C, h) for _ in range(3))
    EM.cell_gates_fwd_planar(X, H, SX, SH, Tc, Wg, bg, U_w, R_w, RH_w)
    nan = lambda *s_: torch.full(s_, float('nan')).cuda()
    U, R, RH = nan(nodes, C, h), nan(nodes, C, h), nan(nodes, C, h)
    hip.cell_gates_fwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(bg), U, R, RH)
    assert rel_err(U, U_w) < TOL and rel_err(R, R_w) < TOL and rel_err(RH, RH_w) < TOL

    dRH, Cand, dHn = rnd(nodes, C, h), torch.tanh(rnd(nodes, C, h)), rnd(nodes, C, h)
    dZ_w = [None, None, torch.empty(nodes, C, h), torch.empty(nodes, C, h)]
    dW_w, db_w, dH_w = torch.empty_like(Wg), torch.empty(2 * h), torch.empty(nodes, C, h)
    EM.cell_gates_bwd_planar(X, H, SX, SH, Tc, Wg, dRH, Cand, U_w, R_w, dHn, dZ_w, dW_w, db_w, dH_w)
    dZ = [None, None, nan(nodes, C, h), nan(nodes, C, h)]
    dW, db, dH = nan(*Wg.shape), nan(2 * h), nan(nodes, C, h)
    hip.cell_gates_bwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(dRH), cu(Cand), cu(U_w), cu(R_w), cu(dHn), dZ, dW, db, dH)
    assert rel_err(dZ[2], dZ_w[2]) < TOL and rel_err(dZ[3], dZ_w[3]) < TOL
    assert rel_err(dW, dW_w) < TOL and rel_err(db, db_w) < TOL and rel_err(dH, dH_w) < TOL
    dZf = [None, None, nan(nodes, C, h), nan(nodes, C, h)]               # dH = None: the share is folded into the H plane's gradient
    hip.cell_gates_bwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(dRH), cu(Cand), cu(U_w), cu(R_w), cu(dHn), dZf, dW, db, None)
    assert rel_err(dZf[2], dZ_w[2] + dH_w) < TOL and torch.equal(dZf[3], dZ[3])

    Wc, bc = rnd(K * K * Lw, h) / (4 * Lw) ** 0.5, rnd(h)
    A_w, B_w = torch.empty(nodes, C, h), torch.empty(nodes, C, h)
    EM.node_post_fwd(RH_w, Tc, Wc, bc, A_w, B_w, X2=X)
    A, Bm = nan(nodes, C, h), nan(nodes, C, h)
    hip.node_post_fwd(cu(RH_w), cu(Tc), cu(Wc), cu(bc), A, Bm, X2=cu(X))
    assert rel_err(A, A_w) < TOL and rel_err(Bm, B_w) < TOL
    if hip.cell_planar_post_fused(C):
        U2, R2, RH2, A2, B2 = (nan(nodes, C, h) for _ in range(5))
        hip.cell_gates_fwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(bg), U2, R2, RH2, post=(cu(Wc), cu(bc), A2, B2))
        assert torch.equal(U2, U) and torch.equal(RH2, RH) and rel_err(A2, A_w) < TOL and rel_err(B2, B_w) < TOL
    dA, dB = rnd(nodes, C, h), rnd(nodes, C, h)
    dX_w, dWc_w, dbc_w = torch.empty(nodes, C, h), torch.empty_like(Wc), torch.empty(h)
    EM.node_post_bwd(RH_w, Tc, Wc, dA, dB, dX_w, dWc_w, dbc_w, X2=X)
    dX, dWc, dbc = nan(nodes, C, h), nan(*Wc.shape), nan(h)
    hip.node_post_bwd(cu(RH_w), cu(Tc), cu(Wc), cu(dA), cu(dB), dX, dWc, dbc, X2=cu(X))
    assert rel_err(dX, dX_w) < TOL and rel_err(dWc, dWc_w) < TOL and rel_err(dbc, dbc_w) < TOL


@pytest.mark.parametrize('nodes,cin,bias', [(50, 16, True), (3, 16, False), (4500, 16, True), (50, 1, True), (9, 3, False), (4500, 1, True), (13, 4, True)])
def test_cell_backward_in_one_launch(hip, nodes, cin, bias):
    """stc_cell_bwd_planar_f32: the candidate's post-aggregation backward + the gates backward of one planar cell step fused per node
    (dY formed inside, R*H formed inside, d(R*H) handed over in the wave, dX = both convolutions' shares) -- against the CPU twin
    (which composes the two separate twins) and against the two separate HIP launches; the forward may skip the R*H plane."""
    C, h, K = 32, 16, 2
    assert hip.cell_bwd_planar_supported(C, h)
    Lw = cin + h
    g = torch.Generator().manual_seed(nodes + cin)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    X, SX, H, SH = rnd(nodes, C, cin), rnd(nodes, C, cin), torch.tanh(rnd(nodes, C, h)), rnd(nodes, C, h)
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    Wg, Wc = rnd(K * K * Lw, 2 * h) / (4 * Lw) ** 0.5, rnd(K * K * Lw, h) / (4 * Lw) ** 0.5
    U, R, Cand = torch.sigmoid(rnd(nodes, C, h)), torch.sigmoid(rnd(nodes, C, h)), torch.tanh(rnd(nodes, C, h))
    dHn, dBm = rnd(nodes, C, h), rnd(nodes, C, h)
    wide = cin == h
    new = lambda dev: [torch.full((nodes, C, h), float('nan'), device=dev) if (wide or i >= 2) else None for i in range(4)]
    dZ_w, dWg_w, dWc_w = new('cpu'), torch.empty_like(Wg), torch.empty_like(Wc)
    dbg_w, dbc_w = (torch.empty(2 * h), torch.empty(h)) if bias else (None, None)
    EM.cell_bwd_planar(X, H, SX, SH, Tc, Wg, Wc, U, R, Cand, dHn, dBm, dZ_w, dWg_w, dbg_w, dWc_w, dbc_w)
    nan = lambda *s_: torch.full(s_, float('nan')).cuda()
    dZ, dWg, dWc = new('cuda'), nan(*Wg.shape), nan(*Wc.shape)
    dbg, dbc = (nan(2 * h), nan(h)) if bias else (None, None)
    ops_ = [cu(t) for t in (X, H, SX, SH, Tc, Wg, Wc, U, R, Cand, dHn, dBm)]
    hip.cell_bwd_planar(*ops_, dZ, dWg, dbg, dWc, dbc)
    for a_, w_ in zip(dZ, dZ_w):
        assert (a_ is None) == (w_ is None) and (a_ is None or rel_err(a_, w_) < TOL)
    assert rel_err(dWg, dWg_w) < TOL and rel_err(dWc, dWc_w) < TOL
    if bias:
        assert rel_err(dbg, dbg_w) < TOL and rel_err(dbc, dbc_w) < TOL
    # the two separate launches on the same operands (dY and R*H as planes)
    dY, RH = cu(dHn * U * (1 - Cand * Cand)), cu(R * H)
    dRH, dXc, dWc2, dZ2, dWg2 = nan(nodes, C, h), nan(nodes, C, h), nan(*Wc.shape), new('cuda'), nan(*Wg.shape)
    if wide:
        hip.node_post_bwd(cu(X), cu(Tc), cu(Wc), dY, cu(dBm), dXc, dWc2, None, X2=RH, dX2=dRH)
    else:
        hip.node_post_bwd(RH, cu(Tc), cu(Wc), dY, cu(dBm), dRH, dWc2, None, X2=cu(X))
    hip.cell_gates_bwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), dRH, cu(Cand), cu(U), cu(R), cu(dHn), dZ2, dWg2, None, None)
    assert rel_err(dWc, dWc2) < 2e-6 and rel_err(dWg, dWg2) < 2e-6 and rel_err(dZ[2], dZ2[2]) < 2e-6 and rel_err(dZ[3], dZ2[3]) < 2e-6
    if wide:
        assert rel_err(dZ[0], dZ2[0] + dXc) < 2e-6 and rel_err(dZ[1], dZ2[1]) < 2e-6
    # accumulate_x / accumulate_h: the planes already hold the state's other consumer's gradients; the launch adds its own
    for acc_x, acc_h in ((wide, False), (False, True), (wide, True)):
        base = [None if z is None else torch.randn(nodes, C, h, generator=g) for z in dZ_w]
        dZa = [None if b_ is None else cu(b_).clone() for b_ in base]
        hip.cell_bwd_planar(*ops_, dZa, dWg, dbg, dWc, dbc, accumulate_x=acc_x, accumulate_h=acc_h)
        for i, (a_, w_, b_) in enumerate(zip(dZa, dZ_w, base)):
            if a_ is not None:
                assert rel_err(a_, w_ + b_ if (acc_x if i < 2 else acc_h) else w_) < TOL, (i, acc_x, acc_h)
        # bitwise reproducible (fixed-order combine of the per-workgroup partial sums)
    dZ3, dWg3, dWc3 = new('cuda'), nan(*Wg.shape), nan(*Wc.shape)
    hip.cell_bwd_planar(*ops_, dZ3, dWg3, None, dWc3, None)
    assert torch.equal(dWg3, dWg) and torch.equal(dWc3, dWc) and torch.equal(dZ3[2], dZ[2])
    # forward with the fused candidate projection and no R*H plane: same U, R, A, Bm as with the plane
    if hip.cell_planar_post_fused(C):
        bg, bc = rnd(2 * h), rnd(h)
        outs = [[nan(nodes, C, h) for _ in range(5)] for _ in range(2)]
        for o, with_rh in zip(outs, (True, False)):
            hip.cell_gates_fwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(bg), o[0], o[1], o[2] if with_rh else None, post=(cu(Wc), cu(bc), o[3], o[4]))
        for i in (0, 1, 3, 4):
            assert torch.equal(outs[0][i], outs[1][i])
        assert torch.isnan(outs[1][2]).all()


@pytest.mark.parametrize('batch,grid,C,n_add,dual', [(2, (5, 5), 32, 3, True), (1, (4, 7), 64, 5, False), (2, (40, 56), 32, 0, True), (1, (1, 1), 32, 2, False)])
def test_state_gradient_from_pieces(hip, batch, grid, C, n_add, dual):
    """stc_spmm_sum_f32: Y = sum of addends (contiguous planes and column slices of wider rows) + S.(X [+ X2])."""
    h = 16
    graph = CsrGraph.queen_grid(*grid, normalize=True)
    n = graph.n
    g = torch.Generator().manual_seed(n + C + n_add)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    X, X2 = rnd(batch, n, C, h), (rnd(batch, n, C, h) if dual else None)
    adds = [(rnd(batch, n, C, 32), 16) if i % 2 else (rnd(batch, n, C, h), 0) for i in range(n_add)]
    hst = graph._host
    csr = tuple(torch.from_numpy(hst[k]) for k in ('bwd_rowptr', 'bwd_colidx', 'bwd_val'))
    dev = graph.on(torch.device('cuda'))
    plan = (dev['bwd_blk_ptr'], dev['bwd_blk_cols'], dev['bwd_blk_vals'])
    Y_w, dY_w = torch.empty(batch, n, C, h), torch.empty(batch, n, C, h)
    U, Cand = torch.rand(batch, n, C, h, generator=g), torch.tanh(rnd(batch, n, C, h))
    EM.spmm_sum(*csr, None, X, X2, adds, Y_w, blend=(U, Cand, dY_w))
    for pl in (plan, None):                                          # row-blocked and plain CSR kernels
        Y = torch.full((batch, n, C, h), float('nan')).cuda()
        hip.spmm_sum(*(cu(t) for t in csr), pl, cu(X), None if X2 is None else cu(X2), [(cu(t), o) for t, o in adds], Y)
        assert rel_err(Y, Y_w) < TOL
        Y2, dY = torch.full_like(Y, float('nan')), torch.full_like(Y, float('nan'))
        hip.spmm_sum(*(cu(t) for t in csr), pl, cu(X), None if X2 is None else cu(X2), [(cu(t), o) for t, o in adds], Y2,
                     blend=(cu(U), cu(Cand), dY))                    # + the owning cell's blend backward in the epilogue
        assert torch.equal(Y2, Y) and rel_err(dY, dY_w) < TOL


@pytest.mark.parametrize('batch,grid,alpha', [(2, (5, 6), 2.0), (1, (30, 41), 2.0), (2, (4, 4), -0.5)])
def test_state_gradient_sum_with_scales(hip, batch, grid, alpha):
    """stc_spmm_sum_f32 with alpha and signed addends: the two launches of the order-3 (Clenshaw) state gradient
    d0 - d2 + S^T (d1 + 2 S^T d2), up to the kernel's eight addends."""
    h, C = 16, 32
    graph = CsrGraph.queen_grid(*grid, normalize=True)
    n = graph.n
    g = torch.Generator().manual_seed(n + 5)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    X, X2 = rnd(batch, n, C, h), rnd(batch, n, C, h)
    adds = [(rnd(batch, n, C, h), 0, sc) for sc in (1.0, 1.0, -1.0, 1.0, -1.0, 0.5, 1.0, -2.0)]
    hst = graph._host
    csr = tuple(torch.from_numpy(hst[k]) for k in ('bwd_rowptr', 'bwd_colidx', 'bwd_val'))
    dev = graph.on(torch.device('cuda'))
    plan = (dev['bwd_blk_ptr'], dev['bwd_blk_cols'], dev['bwd_blk_vals'])
    for n_add in (8, 3, 0):
        Y_w = torch.empty(batch, n, C, h)
        EM.spmm_sum(*csr, None, X, X2, adds[:n_add], Y_w, alpha=alpha)
        want = alpha * torch.einsum('ij,bjch->bich', graph.to_dense(), X + X2) + sum((sc * t for t, _, sc in adds[:n_add]), torch.zeros(()))
        assert rel_err(Y_w, want) < 1e-6                                 # the twin itself against plain dense algebra
        for pl in (plan, None):
            Y = torch.full((batch, n, C, h), float('nan')).cuda()
            hip.spmm_sum(*(cu(t) for t in csr), pl, cu(X), cu(X2), [(cu(t), o, sc) for t, o, sc in adds[:n_add]], Y, alpha=alpha)
            assert rel_err(Y, Y_w) < TOL
    from stc_hip import StcError
    with pytest.raises(StcError):
        hip.spmm_sum(*(cu(t) for t in csr), plan, cu(X), None, [(cu(X2), 0)] * 9, torch.empty_like(cu(X)))


@pytest.mark.parametrize('nodes,cin', [(50, 16), (4500, 16), (37, 1), (600, 4), (9, 3)])
def test_planar_cell_kernels_order3(hip, nodes, cin):
    """Order-3 planar cell kernels (stc_cell_{gates,cand}_{fwd,bwd}_planar_k_f32): three Chebyshev planes per side, 16 + 16
    columns or a narrow input plane, against the CPU twin (slab form on the concatenated planes)."""
    h, K, C = 16, 3, 32
    assert hip.cell_planar_k_supported(K, C, h) and not hip.cell_planar_k_supported(2, C, h) and not hip.cell_planar_k_supported(3, 64, h)
    Lw = cin + h
    g = torch.Generator().manual_seed(nodes + cin)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    Zx, Zh, Zr = [rnd(nodes, C, cin) for _ in range(K)], [rnd(nodes, C, h) for _ in range(K)], [rnd(nodes, C, h) for _ in range(K)]
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    Wg, bg = rnd(K * K * Lw, 2 * h) / (K * K * Lw) ** 0.5, rnd(2 * h)
    Wc, bc = rnd(K * K * Lw, h) / (K * K * Lw) ** 0.5, rnd(h)
    nan = lambda *s_: torch.full(s_, float('nan')).cuda()
    cus = lambda ts: [cu(t) for t in ts]
    # gates forward
    U_w, R_w, RH_w = (torch.empty(nodes, C, h) for _ in range(3))
    EM.cell_gates_fwd_planar_k(Zx, Zh, Tc, Wg, bg, U_w, R_w, RH_w)
    U, R, RH = nan(nodes, C, h), nan(nodes, C, h), nan(nodes, C, h)
    hip.cell_gates_fwd_planar_k(cus(Zx), cus(Zh), cu(Tc), cu(Wg), cu(bg), U, R, RH)
    assert rel_err(U, U_w) < TOL and rel_err(R, R_w) < TOL and rel_err(RH, RH_w) < TOL
    # candidate forward + blend
    Cand_w, Hn_w = torch.empty(nodes, C, h), torch.empty(nodes, C, h)
    EM.cell_cand_fwd_planar_k(Zx, Zr, Tc, Wc, bc, U_w, Zh[0], Cand_w, Hn_w)
    Cand, Hn = nan(nodes, C, h), nan(nodes, C, h)
    hip.cell_cand_fwd_planar_k(cus(Zx), cus(Zr), cu(Tc), cu(Wc), cu(bc), cu(U_w), cu(Zh[0]), Cand, Hn)
    assert rel_err(Cand, Cand_w) < TOL and rel_err(Hn, Hn_w) < TOL
    hip.cell_cand_fwd_planar_k(cus(Zx), cus(Zr), cu(Tc), cu(Wc), None, cu(U_w), cu(Zh[0]), Cand, Hn)        # no bias
    EM.cell_cand_fwd_planar_k(Zx, Zr, Tc, Wc, None, U_w, Zh[0], Cand_w, Hn_w)
    assert rel_err(Cand, Cand_w) < TOL
    EM.cell_cand_fwd_planar_k(Zx, Zr, Tc, Wc, bc, U_w, Zh[0], Cand_w, Hn_w)
    # candidate backward (blend backward in the prologue)
    wide = cin == h
    dHn = rnd(nodes, C, h)
    gx = lambda make: [make() for _ in range(K)] if wide else [None] * K
    dXc_w, dR_w = gx(lambda: torch.empty(nodes, C, cin)), [torch.empty(nodes, C, h) for _ in range(K)]
    dWc_w, dbc_w = torch.empty_like(Wc), torch.empty(h)
    EM.cell_cand_bwd_planar_k(Zx, Zr, Tc, Wc, dHn, U_w, Cand_w, dXc_w, dR_w, dWc_w, dbc_w)
    dXc, dR = gx(lambda: nan(nodes, C, cin)), [nan(nodes, C, h) for _ in range(K)]
    dWc, dbc = nan(*Wc.shape), nan(h)
    hip.cell_cand_bwd_planar_k(cus(Zx), cus(Zr), cu(Tc), cu(Wc), cu(dHn), cu(U_w), cu(Cand_w), dXc, dR, dWc, dbc)
    for a, w in zip(dR + (dXc if wide else []), dR_w + (dXc_w if wide else [])):
        assert rel_err(a, w) < TOL
    assert rel_err(dWc, dWc_w) < TOL and rel_err(dbc, dbc_w) < TOL
    # gates backward (gate + blend backward in the prologue)
    dRH = rnd(nodes, C, h)
    dXg_w, dHg_w = gx(lambda: torch.empty(nodes, C, cin)), [torch.empty(nodes, C, h) for _ in range(K)]
    dWg_w, dbg_w, dH_w = torch.empty_like(Wg), torch.empty(2 * h), torch.empty(nodes, C, h)
    EM.cell_gates_bwd_planar_k(Zx, Zh, Tc, Wg, dRH, Cand_w, U_w, R_w, dHn, dXg_w, dHg_w, dWg_w, dbg_w, dH_w)
    dXg, dHg = gx(lambda: nan(nodes, C, cin)), [nan(nodes, C, h) for _ in range(K)]
    dWg, dbg, dH = nan(*Wg.shape), nan(2 * h), nan(nodes, C, h)
    hip.cell_gates_bwd_planar_k(cus(Zx), cus(Zh), cu(Tc), cu(Wg), cu(dRH), cu(Cand_w), cu(U_w), cu(R_w), cu(dHn), dXg, dHg, dWg, dbg, dH)
    for a, w in zip(dHg + (dXg if wide else []), dHg_w + (dXg_w if wide else [])):
        assert rel_err(a, w) < TOL
    assert rel_err(dWg, dWg_w) < TOL and rel_err(dbg, dbg_w) < TOL and rel_err(dH, dH_w) < TOL
    dHf = [nan(nodes, C, h) for _ in range(K)]                            # dH = None: folded into dZh[0]
    hip.cell_gates_bwd_planar_k(cus(Zx), cus(Zh), cu(Tc), cu(Wg), cu(dRH), cu(Cand_w), cu(U_w), cu(R_w), cu(dHn), dXg, dHf, dWg, dbg, None)
    assert rel_err(dHf[0], dHg_w[0] + dH_w) < TOL and torch.equal(dHf[1], dHg[1]) and torch.equal(dHf[2], dHg[2])
    if wide:                                                              # accumulate_x: the gates' X-side gradients are added into planes that hold the candidate's
        acc = [t.clone() for t in dXc]
        hip.cell_gates_bwd_planar_k(cus(Zx), cus(Zh), cu(Tc), cu(Wg), cu(dRH), cu(Cand_w), cu(U_w), cu(R_w), cu(dHn), acc, dHf, dWg, dbg, None, accumulate_x=True)
        for n in range(K):
            assert rel_err(acc[n], dXc_w[n] + dXg_w[n]) < TOL
        assert rel_err(dWg, dWg_w) < TOL


@pytest.mark.parametrize('nodes,C,cin,K', [(50, 32, 16, 2), (50, 32, 1, 2), (21, 16, 16, 3), (13, 64, 1, 2), (4500, 32, 16, 2), (9, 32, 13, 1)])
def test_fused_gates_backward_prologue(hip, nodes, C, cin, K, fused_path):
    """Gate backward as the prologue of the node backward (dG never stored) vs gate kernel + node backward of the twin."""
    h = 16
    Lw = cin + h
    L = Lw + (-Lw) % 4
    g = torch.Generator().manual_seed(nodes + C + cin + K + 1)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    Zs = [rnd(nodes, C, L) for _ in range(K)]
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    W = rnd(K * K * Lw, 2 * h) / (K * K * Lw) ** 0.5
    dCand, dU, H = rnd(nodes, C, L), rnd(nodes, C, h), rnd(nodes, C, h)
    U, R, owed = torch.rand(nodes, C, h, generator=g), torch.rand(nodes, C, h, generator=g), rnd(nodes, C, h)
    dZ_w = [torch.empty(nodes, C, L) for _ in range(K)]
    dW_w, db_w, dXt_w, dH_w = torch.empty_like(W), torch.empty(2 * h), torch.empty(nodes, C, cin), owed.clone()
    EM.cell_gates_bwd(Zs, Tc, W, dCand, dU, H, U, R, dH_w, dZ_w, dW_w, db_w, dXt_w, dH_w)
    nan = float('nan')
    dZ = [torch.full((nodes, C, L), nan).cuda() for _ in range(K)]
    dW, db = torch.full_like(W, nan).cuda(), torch.full((2 * h,), nan).cuda()
    dXt, dH = torch.full((nodes, C, cin), nan).cuda(), owed.clone().cuda()
    hip.cell_gates_bwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(dCand), cu(dU), cu(H), cu(U), cu(R), dH, dZ, dW, db, dXt, dH)
    for a, w in zip(dZ, dZ_w):
        assert rel_err(a, w) < TOL
    assert rel_err(dW, dW_w) < TOL and rel_err(db, db_w) < TOL
    assert torch.equal(dXt.cpu(), dCand[..., :cin]) and rel_err(dH, dH_w) < TOL
    dH2 = torch.full((nodes, C, h), nan).cuda()                  # nothing owed yet: dH_in = NULL
    hip.cell_gates_bwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(dCand), cu(dU), cu(H), cu(U), cu(R), None, dZ, dW, None, dXt, dH2)
    assert rel_err(dH2, dH_w - owed) < TOL
    # dXt not wanted (read in place by the caller) and dH_in = gradient of the new state, taken times (1 - U) inside
    dH3 = torch.full((nodes, C, h), nan).cuda()
    hip.cell_gates_bwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(dCand), cu(dU), cu(H), cu(U), cu(R), cu(owed), dZ, dW, None, None, dH3,
                       dH_in_scaled=True)
    assert rel_err(dH3, dH_w - owed + owed * (1 - U)) < TOL
    for a, w in zip(dZ, dZ_w):
        assert rel_err(a, w) < TOL
    # Cand form: dH_in is dHnew; dU = dHnew * (Cand - H) and the state share are formed inside (no blend backward pass)
    Cand = torch.tanh(rnd(nodes, C, h))
    dH_w4 = torch.empty(nodes, C, h)
    EM.cell_gates_bwd(Zs, Tc, W, dCand, None, H, U, R, owed, dZ_w, dW_w, db_w, None, dH_w4, Cand=Cand)
    dH4 = torch.full((nodes, C, h), nan).cuda()
    hip.cell_gates_bwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(dCand), None, cu(H), cu(U), cu(R), cu(owed), dZ, dW, db, None, dH4,
                       dH_in_scaled=True, Cand=cu(Cand))
    for a, w in zip(dZ, dZ_w):
        assert rel_err(a, w) < TOL
    assert rel_err(dW, dW_w) < TOL and rel_err(db, db_w) < TOL and rel_err(dH4, dH_w4) < TOL


@pytest.mark.parametrize('nodes,C,cin,K', [(50, 32, 16, 2), (50, 32, 1, 2), (21, 16, 16, 3), (13, 64, 1, 2), (4500, 32, 16, 2), (9, 32, 13, 1)])
def test_fused_candidate_backward_prologue(hip, nodes, C, cin, K, fused_path):
    """Blend backward as the prologue of the candidate convolution's node backward vs blend kernel + node backward of the twin."""
    h = 16
    Lw = cin + h
    L = Lw + (-Lw) % 4
    g = torch.Generator().manual_seed(nodes + C + cin + K + 2)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    Zs = [rnd(nodes, C, L) for _ in range(K)]
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    W = rnd(K * K * Lw, h) / (K * K * Lw) ** 0.5
    dHn, U, Cand = rnd(nodes, C, h), torch.rand(nodes, C, h, generator=g), torch.tanh(rnd(nodes, C, h))
    dZ_w = [torch.empty(nodes, C, L) for _ in range(K)]
    dW_w, db_w = torch.empty_like(W), torch.empty(h)
    EM.cell_cand_bwd(Zs, Tc, W, dHn, U, Cand, dZ_w, dW_w, db_w)
    nan = float('nan')
    dZ = [torch.full((nodes, C, L), nan).cuda() for _ in range(K)]
    dW, db = torch.full_like(W, nan).cuda(), torch.full((h,), nan).cuda()
    hip.cell_cand_bwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(dHn), cu(U), cu(Cand), dZ, dW, db)
    for a, w in zip(dZ, dZ_w):
        assert rel_err(a, w) < TOL
    assert rel_err(dW, dW_w) < TOL and rel_err(db, db_w) < TOL
    hip.cell_cand_bwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(dHn), cu(U), cu(Cand), dZ, dW, None)      # convolution without bias
    assert rel_err(dW, dW_w) < TOL


def test_fused_cell_unsupported_shapes_are_refused(hip, monkeypatch):
    from stc_hip._lib import StcError
    assert not hip.cell_fused_supported(2, 2, 5, 20, 16)          # SF category count
    assert not hip.cell_fused_supported(2, 2, 32, 32, 8)          # hidden != 16
    assert not hip.cell_fused_supported(2, 3, 32, 32, 16)         # Ks != Kc
    Zs = [torch.randn(6, 5, 20).cuda() for _ in range(2)]
    Tc = torch.eye(5).repeat(2, 1, 1).cuda()
    H = torch.randn(6, 5, 16).cuda()
    with pytest.raises(StcError, match='fused path'):
        hip.cell_gates_fwd(Zs, Tc, torch.randn(2 * 2 * 17, 32).cuda(), None, H, torch.empty_like(H), torch.empty_like(H),
                           torch.empty(6, 5, 20).cuda())
    hip.set_dispatch_level(2)
    try:
        assert not hip.cell_fused_supported(2, 2, 32, 32, 16)
        with pytest.raises(StcError, match='level'):
            hip.set_dispatch_level(3)
    finally:
        hip.set_dispatch_level(0)


def test_bdg_node_bwd_many_tiles_exercises_grid_stride(hip):
    shape = (3000, 8, 9, 5, 2, 2)                               # 750 tiles > 512 workgroups
    nodes, C, L, Ho, Ks, Kc = shape
    Zs, Tc, W, b, dY = _node_inputs(*shape, seed=5)
    dZ_w = [torch.empty(nodes, C, L) for _ in range(Ks)]
    dW_w, db_w, dT_w = torch.empty_like(W), torch.empty(Ho), torch.empty_like(Tc)
    EM.bdg_node_bwd(Zs, Tc, W, dY, dZ_w, dW_w, db_w, dT_w)
    dZ = [torch.empty(nodes, C, L).cuda() for _ in range(Ks)]
    dW, db, dT = torch.empty_like(W).cuda(), torch.empty(Ho).cuda(), torch.empty_like(Tc).cuda()
    hip.bdg_node_bwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(dY), dZ, dW, db, dT)
    assert rel_err(dZ[1], dZ_w[1]) < TOL and rel_err(dW, dW_w) < TOL and rel_err(db, db_w) < TOL
    assert rel_err(dT[1], dT_w[1]) < TOL
    Y_w = torch.empty(nodes, C, Ho)
    EM.bdg_node_fwd(Zs, Tc, W, b, Y_w)
    Y = torch.empty(nodes, C, Ho).cuda()
    hip.bdg_node_fwd([cu(z) for z in Zs], cu(Tc), cu(W), cu(b), Y)
    assert rel_err(Y, Y_w) < TOL


@pytest.mark.parametrize('rows_shape,cin,h,pad', [((2, 7, 3), 1, 4, 0), ((3, 50, 5), 16, 16, 0), ((1, 9, 2), 4, 5, 3),
                                                ((2, 11, 32), 1, 16, 3)])
def test_gru_gates_and_blend(hip, rows_shape, cin, h, pad):
    g = torch.Generator().manual_seed(cin * 100 + h)
    G = torch.randn(*rows_shape, 2 * h, generator=g)
    Xt = torch.randn(*rows_shape, cin, generator=g)
    H = torch.randn(*rows_shape, h, generator=g)
    U_w, R_w, Ci_w = torch.empty_like(H), torch.empty_like(H), torch.full((*rows_shape, cin + h + pad), 3.0)
    EM.gru_gates_fwd(G, Xt, H, U_w, R_w, Ci_w)
    U, R, Ci = torch.empty_like(H).cuda(), torch.empty_like(H).cuda(), torch.full((*rows_shape, cin + h + pad), 5.0).cuda()
    hip.gru_gates_fwd(cu(G), cu(Xt), cu(H), U, R, Ci)
    assert rel_err(U, U_w) < TOL and rel_err(R, R_w) < TOL and rel_err(Ci, Ci_w) < TOL
    dCi, dU = torch.randn(*rows_shape, cin + h + pad, generator=g), torch.randn(*rows_shape, h, generator=g)
    dG_w, dX_w, dH_w = torch.empty_like(G), torch.empty_like(Xt), torch.empty_like(H)
    EM.gru_gates_bwd(dCi, dU, H, U_w, R_w, dG_w, dX_w, dH_w)
    dG, dX, dH = torch.empty_like(G).cuda(), torch.empty_like(Xt).cuda(), torch.empty_like(H).cuda()
    hip.gru_gates_bwd(cu(dCi), cu(dU), cu(H), cu(U_w), cu(R_w), dG, dX, dH)
    assert rel_err(dG, dG_w) < TOL and rel_err(dX, dX_w) < TOL and rel_err(dH, dH_w) < TOL
    owed = torch.randn(*rows_shape, h, generator=g)                 # a gradient already owed to H, accumulated in place
    dH2 = owed.clone().cuda()
    hip.gru_gates_bwd(cu(dCi), cu(dU), cu(H), cu(U_w), cu(R_w), dG, dX, dH2, dH_in=dH2)
    assert rel_err(dH2, dH_w + owed) < TOL

    Cpre = torch.randn(*rows_shape, h, generator=g) * 2
    Cand_w, Hn_w = torch.empty_like(H), torch.empty_like(H)
    EM.gru_blend_fwd(Cpre, U_w, H, Cand_w, Hn_w)
    Cand, Hn = torch.empty_like(H).cuda(), torch.empty_like(H).cuda()
    hip.gru_blend_fwd(cu(Cpre), cu(U_w), cu(H), Cand, Hn)
    assert rel_err(Cand, Cand_w) < TOL and rel_err(Hn, Hn_w) < TOL
    dHn = torch.randn(*rows_shape, h, generator=g)
    outs_w = [torch.empty_like(H) for _ in range(3)]
    EM.gru_blend_bwd(dHn, U_w, H, Cand_w, *outs_w)
    outs = [torch.empty_like(H).cuda() for _ in range(3)]
    hip.gru_blend_bwd(cu(dHn), cu(U_w), cu(H), cu(Cand_w), *outs)
    for a, w in zip(outs, outs_w):
        assert rel_err(a, w) < TOL
    keep = outs[2].clone()
    hip.gru_blend_bwd(cu(dHn), cu(U_w), cu(H), cu(Cand_w), outs[0], outs[1], None)    # the state's share not wanted
    assert torch.equal(outs[2], keep) and rel_err(outs[0], outs_w[0]) < TOL


@pytest.mark.parametrize('shape,h', [((2, 3, 50, 5), 16), ((1, 6, 777, 4), 8), ((3, 7), 64), ((1, 300000), 16)])
def test_output_head(hip, shape, h):
    g = torch.Generator().manual_seed(h + len(shape))
    H = torch.randn(*shape, h, generator=g)
    w, b = torch.randn(h, generator=g) * 0.5, torch.randn(1, generator=g)
    y_w = torch.empty(*shape)
    EM.head_fwd(H, w, b, y_w)
    y = torch.empty(*shape).cuda()
    hip.head_fwd(cu(H), cu(w), cu(b), y)
    assert rel_err(y, y_w) < TOL
    dy = torch.randn(*shape, generator=g)
    dH_w, dwb_w = torch.empty_like(H), torch.empty(h + 1)
    EM.head_bwd(H, w, y_w, dy, dH_w, dwb_w)
    dH, dwb = torch.empty_like(H).cuda(), torch.empty(h + 1).cuda()
    hip.head_bwd(cu(H), cu(w), cu(y_w), cu(dy), dH, dwb)
    assert rel_err(dH, dH_w) < TOL and rel_err(dwb, dwb_w) < 2e-5
    dwb2 = torch.empty(h + 1).cuda()
    hip.head_bwd(cu(H), cu(w), cu(y_w), cu(dy), dH, dwb2)
    assert torch.equal(dwb, dwb2)                                  # fixed-order reduction


def test_axpy_concat_split(hip):
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    yd = y.clone().cuda()
    hip.axpy(-1.0, cu(x), yd)
    assert rel_err(yd, y - x) < 1e-7
    A, Bm = torch.randn(4, 9, 3, 1, generator=g), torch.randn(4, 9, 3, 16, generator=g)
    out = torch.empty(4, 9, 3, 17).cuda()
    hip.concat2(cu(A), cu(Bm), out)
    assert torch.equal(out.cpu(), torch.cat([A, Bm], -1))
    A2, B2 = torch.empty_like(A).cuda(), torch.empty_like(Bm).cuda()
    hip.split2(out, A2, B2)
    assert torch.equal(A2.cpu(), A) and torch.equal(B2.cpu(), Bm)
    padded = torch.full((4, 9, 3, 20), 9.0).cuda()                # 1 + 16 -> 20: three zero pad columns
    hip.concat2(cu(A), cu(Bm), padded)
    assert torch.equal(padded.cpu(), torch.cat([A, Bm, torch.zeros(4, 9, 3, 3)], -1))
    hip.split2(padded, A2, B2)
    assert torch.equal(A2.cpu(), A) and torch.equal(B2.cpu(), Bm)
    hip.split2(padded, A2, B2, addA=A2, addB=B2)                  # accumulate into existing gradients, in place
    assert torch.equal(A2.cpu(), 2 * A) and torch.equal(B2.cpu(), 2 * Bm)
    A3, B3 = torch.randn(4, 9, 3, 4, generator=g), torch.randn(4, 9, 3, 16, generator=g)   # 16-byte path
    whole = torch.empty(4, 9, 3, 20).cuda()
    hip.concat2(cu(A3), cu(B3), whole)
    oa, ob = torch.ones(4, 9, 3, 4).cuda(), torch.ones(4, 9, 3, 16).cuda()
    hip.split2(whole, oa, ob, addA=oa, addB=None)
    assert torch.equal(oa.cpu(), A3 + 1) and torch.equal(ob.cpu(), B3)
    # addA read in place from the first columns of a wider buffer (row stride addA_ld): 16-byte and scalar paths
    wide = torch.randn(4, 9, 3, 20, generator=g)
    hip.split2(whole, oa, ob, addA=cu(wide), addB=None, addA_ld=20)
    assert torch.equal(oa.cpu(), A3 + wide[..., :4]) and torch.equal(ob.cpu(), B3)
    hip.split2(padded, A2, B2, addA=cu(wide), addB=B2, addA_ld=20)
    assert torch.equal(A2.cpu(), A + wide[..., :1])


# ------------------------------------------------------------------ full-size properties (N = 50 176)
def test_full_size_spmm_properties(hip):
    """BASELINE metric size (224x224 queen grid, C=32, L=32 -> F=1024): properties that need no oracle.

    * CSR(Gs) of the row-stochastic grid maps the all-ones field to itself;
    * adjoint identity <Gs^T x, y> = <x, Gs y> ties the forward and backward operands together;
    * linearity in X;
    * a randomly permuted node order gives the permuted result.
    """
    H = W = 224
    graph = CsrGraph.queen_grid(H, W, normalize=True)
    d = graph.on(torch.device('cuda'))
    N, F = H * W, 1024
    assert graph.nnz == 398724 and 4.4 < graph.fetches_per_row[0] < 4.6
    plan_f = (d['fwd_blk_ptr'], d['fwd_blk_cols'], d['fwd_blk_vals'])
    plan_b = (d['bwd_blk_ptr'], d['bwd_blk_cols'], d['bwd_blk_vals'])
    ones = torch.ones(1, N, F, device='cuda')
    out = torch.empty_like(ones)
    hip.csr_spmm(d['bwd_rowptr'], d['bwd_colidx'], d['bwd_val'], N, N, ones, None, out, 1.0, 0.0, plan=plan_b)
    assert float((out - 1).abs().max()) < 1e-6
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(1, N, F, device='cuda', generator=g)
    y = torch.randn(1, N, F, device='cuda', generator=g)
    STx, Sy = torch.empty_like(x), torch.empty_like(x)
    hip.csr_spmm(d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'], N, N, x, None, STx, 1.0, 0.0, plan=plan_f)   # row-blocked
    hip.csr_spmm(d['bwd_rowptr'], d['bwd_colidx'], d['bwd_val'], N, N, y, None, Sy, 1.0, 0.0)                # direct
    lhs, rhs = (STx.double() * y.double()).sum(), (x.double() * Sy.double()).sum()
    assert abs(float(lhs - rhs)) < 1e-6 * abs(float(rhs)) + 1e-3
    both = torch.empty_like(x)
    hip.csr_spmm(d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'], N, N, 2 * x + y, None, both, 1.0, 0.0)
    STy = torch.empty_like(x)
    hip.csr_spmm(d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'], N, N, y, None, STy, 1.0, 0.0)
    assert rel_err(both, 2 * STx + STy) < TOL
    # permuted node order (seed 1234): P-conjugated graph on P-permuted features = permuted result
    perm_graph = CsrGraph.queen_grid(H, W, normalize=True, permute_seed=1234)
    p = torch.randperm(N, generator=torch.Generator().manual_seed(1234)).cuda()
    dp = perm_graph.on(torch.device('cuda'))
    out_p = torch.empty_like(x)
    hip.csr_spmm(dp['fwd_rowptr'], dp['fwd_colidx'], dp['fwd_val'], N, N, x[:, p].contiguous(), None, out_p, 1.0, 0.0)
    assert rel_err(out_p, STx[:, p]) < TOL


@pytest.mark.parametrize('n,needs_dA', [(10, True), (14, False), (32, True), (100, False)])
def test_mixed_fusion(hip, n, needs_dA):
    """stc_mixed_fusion_fwd/bwd_f32 (reference STC_GNN.py:246-261) against float64 autograd of the reference's own expression: the mixed graph,
    and the gradients of both weight matrices, both biases, P and (when asked for) A; D = n^2 in {100, 196, 1024, 10^4 (the SF shape)}."""
    from stc_hip import ops as O2
    D = n * n
    g = torch.Generator().manual_seed(n)
    A = torch.rand(n, n, generator=g)
    P = torch.softmax(torch.randn(n, n, generator=g), -1)
    WA, WP = (torch.randn(D, D, generator=g) * (1.0 / D) ** 0.5 for _ in range(2))
    bA, bP = (torch.randn(D, generator=g) * 0.1 for _ in range(2))
    R = torch.randn(n, n, generator=g)
    ref = [t.double().requires_grad_() for t in (A, P, WA, bA, WP, bP)]
    a = torch.sigmoid(ref[2] @ ref[0].reshape(D) + ref[3] + ref[4] @ ref[1].reshape(D) + ref[5]).reshape(n, n)
    Gref = a * ref[0] + (1 - a) * ref[1]
    (Gref * R.double()).sum().backward()
    dev = [t.cuda().requires_grad_(i != 0 or needs_dA) for i, t in enumerate((A, P, WA, bA, WP, bP))]
    assert O2.mixed_fusion_supported(*dev)
    G = O2.mixed_fusion(*dev)
    (G * R.cuda()).sum().backward()
    assert rel_err(G, Gref) < TOL
    for name, got, want in zip(('dA', 'dP', 'dWA', 'dbA', 'dWP', 'dbP'), dev, ref):
        if name == 'dA' and not needs_dA:
            assert got.grad is None
            continue
        assert rel_err(got.grad, want.grad) < 2e-5, name
    # frozen weights: the two (n^2, n^2) gradients are neither allocated nor written; the other gradients are unchanged, bit for bit;
    # an operand that is not 16-byte aligned (a view into a flat buffer) is refused by the predicate, not by the kernel
    frozen = [t.detach().clone().requires_grad_(i in (1, 3, 5) or (i == 0 and needs_dA)) for i, t in enumerate(dev)]
    (O2.mixed_fusion(*frozen) * R.cuda()).sum().backward()
    assert frozen[2].grad is None and frozen[4].grad is None
    assert torch.equal(frozen[1].grad, dev[1].grad) and torch.equal(frozen[3].grad, dev[3].grad)
    flat = torch.zeros(WA.numel() + 1, device='cuda')
    assert not O2.mixed_fusion_supported(dev[0], dev[1], flat[1:].view_as(WA), dev[3], dev[4], dev[5])


@pytest.mark.parametrize('nodes,C,L,Lw,Ho,K', [
    (6400, 8, 32, 32, 32, 2),    # BASELINE configuration 2: 3 200 tiles, more than the grid's waves
    (6400, 8, 32, 32, 32, 3),
    (6, 8, 20, 17, 32, 2),       # padded rows (in + hidden = 17), three tiles
    (40, 4, 20, 18, 16, 3),
    (16, 2, 32, 32, 16, 2),
    (48, 1, 20, 17, 32, 3),
    (5, 16, 32, 32, 32, 2),      # C = 16: one node per tile
    (2, 8, 32, 32, 16, 1),       # K = 1: T_0 = I only, zeros
    (3000, 5, 20, 17, 32, 3),    # the SF shape's C = 5: three nodes = 15 rows per tile
    (33, 5, 32, 32, 16, 2),
    (14, 7, 32, 32, 32, 2),      # two nodes = 14 rows per tile
    (9, 11, 20, 18, 16, 2),      # one node of 11 rows per tile
    (3200, 5, 32, 32, 16, 3),    # 3 200 nodes in tiles of three: the last tile holds two
    (7, 8, 32, 32, 32, 2),       # an odd node count at two nodes per tile
])
def test_mix_dT_for_few_categories(hip, nodes, C, L, Lw, Ho, K):
    """stc_mix_dt_f32 (the category graph's gradient the packed matrix-core node backward leaves to its caller) against its twin, and the
    same number twice (fixed-order sums)."""
    g = torch.Generator().manual_seed(nodes + 10 * C + K)
    Zs = [torch.randn(nodes, C, L, generator=g) for _ in range(K)]
    for z in Zs:
        z[..., Lw:] = 7.0                                          # pad columns hold anything: W has no rows for them
    W = torch.randn(K * K * Lw, Ho, generator=g)
    dY = torch.randn(nodes, C, Ho, generator=g)
    assert hip.mix_dT_supported(K, K, C, L, Ho) and EM.mix_dT_supported(K, K, C, L, Ho)
    want = torch.empty(K, C, C, dtype=torch.float64)               # (sums over up to 2e5 products: the float64 twin is the exact side)
    EM.mix_dT([z.double() for z in Zs], W.double(), dY.double(), want)
    got = torch.full((K, C, C), float('nan')).cuda()
    hip.mix_dT([cu(z) for z in Zs], cu(W), cu(dY), got)
    assert float(got[0].abs().max()) == 0.0                        # T_0 = I is a constant
    if K > 1:
        assert rel_err(got[1:], want[1:]) < TOL
    again = torch.empty_like(got)
    hip.mix_dT([cu(z) for z in Zs], cu(W), cu(dY), again)
    assert torch.equal(got, again)


def test_mix_dT_rejects_what_it_does_not_take(hip):
    from stc_hip._lib import StcError
    assert not hip.mix_dT_supported(2, 2, 17, 32, 32) and not hip.mix_dT_supported(2, 3, 8, 32, 32) and not hip.mix_dT_supported(2, 2, 8, 24, 32)
    with pytest.raises(StcError, match='unsupported|Ks = Kc'):
        hip.mix_dT([torch.zeros(4, 17, 32).cuda() for _ in range(2)], torch.zeros(128, 32).cuda(), torch.zeros(4, 17, 32).cuda(), torch.zeros(2, 17, 17).cuda())
