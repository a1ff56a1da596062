"""-m gpu: the bf16-storage kernels (BASELINE.json configuration 5) through the C ABI.

The reference has no bf16 behaviour, so the contract is the build's own: a bf16-storage kernel equals the
fp32 kernel run on the same bf16-valued inputs, rounded to bf16 once.  fp32 sums taken in a different order can land on
the other side of a rounding boundary, so the bound is ONE bf16 ulp per element (2^-8 relative, checked as
|a - b| <= 2^-7 |b| + the fp32 summation noise of the terms), with all but a small fraction of the elements bit-identical.
"""
import pytest
import torch

from oracle.kernel_emul import EmulatedKernels
from stc_hip import CsrGraph
from tests.conftest import rel_err
from tests.test_hip_kernels import _banded_graph, cu, random_csr

pytestmark = pytest.mark.gpu
EM = EmulatedKernels()


@pytest.fixture(scope='module')
def hip():
    from stc_hip._lib import HipKernels
    return HipKernels()


def assert_one_ulp(got, want, max_mismatch=0.02):
    """bf16 tensors: every element within one bf16 ulp, at most ``max_mismatch`` of them different at all."""
    g, w = got.float().cpu(), want.float().cpu()
    assert torch.isfinite(g).all()
    diff = (g - w).abs()
    # a result that cancels to far below its terms carries the fp32 summation-order noise of those terms, which can
    # exceed a bf16 ulp of the tiny result: allow that noise (4e-6 of the largest value) besides the one ulp
    floor = 4e-6 * float(w.abs().max())
    assert bool((diff <= w.abs() * 2.0 ** -7 + floor).all()), f'beyond one bf16 ulp: max diff {float(diff.max())}'
    assert float((diff > 0).float().mean()) <= max_mismatch


@pytest.mark.parametrize('n_rows,n_cols,F,B,density', [
    (37, 37, 1024, 2, 0.2),      # two pieces per lane, ragged last tile
    (64, 50, 160, 3, 0.3),       # 20 pieces: lanes beyond the row idle
    (40, 40, 512, 1, 1.0),       # exactly one piece per lane
    (300, 300, 256, 1, 1.0),     # rows longer than the staged segment
    (9, 9, 4096, 1, 0.5),        # several column blocks per row
    (5, 7, 8, 1, 0.5),           # one piece per row
])
@pytest.mark.parametrize('alpha,beta', [(1.0, 0.0), (2.0, -1.0)])
def test_csr_spmm_bf16(hip, n_rows, n_cols, F, B, density, alpha, beta):
    rowptr, colidx, val, dense = random_csr(n_rows, n_cols, density, seed=n_rows * 7 + F, empty_rows=(1, n_rows - 1))
    g = torch.Generator().manual_seed(F)
    X = torch.randn(B, n_cols, F, generator=g).bfloat16()
    Y0 = torch.randn(B, n_rows, F, generator=g).bfloat16() if beta != 0 else None
    want = torch.empty(B, n_rows, F, dtype=torch.bfloat16)
    EM.csr_spmm_bf16(rowptr, colidx, val, n_rows, n_cols, X, Y0, want, alpha, beta)
    got = torch.full((B, n_rows, F), float('nan'), dtype=torch.bfloat16).cuda()
    hip.csr_spmm_bf16(cu(rowptr), cu(colidx), cu(val), n_rows, n_cols, cu(X), cu(Y0), got, alpha, beta)
    assert_one_ulp(got, want)
    # the fp32 kernel on the same bf16-valued inputs, rounded once
    f32 = torch.empty(B, n_rows, F).cuda()
    hip.csr_spmm(cu(rowptr), cu(colidx), cu(val), n_rows, n_cols, cu(X.float()), None if Y0 is None else cu(Y0.float()), f32, alpha, beta)
    assert_one_ulp(got, f32.bfloat16())
    if Y0 is not None:                                          # in-place epilogue: Y0 aliases Y
        buf = Y0.clone().cuda()
        hip.csr_spmm_bf16(cu(rowptr), cu(colidx), cu(val), n_rows, n_cols, cu(X), buf, buf, alpha, beta)
        assert torch.equal(buf, got)


@pytest.mark.parametrize('n,F,B,hw', [(203, 2048, 1, 4), (77, 640, 2, 3), (64, 256, 1, 6), (1001, 64, 1, 2), (30, 4096, 1, 40)])
def test_bcsr_spmm_bf16_equals_csr(hip, n, F, B, hw):
    """Row-blocked bf16 kernel vs the CSR bf16 kernel vs the fp32 product: ragged last block, an empty row, block lists
    longer than the staged segment (hw = 40), several column blocks, in-place beta epilogue, both graph orientations."""
    graph, V = _banded_graph(n, hw, seed=n + F)
    d = graph.on(torch.device('cuda'))
    g = torch.Generator().manual_seed(F)
    X = torch.randn(B, n, F, generator=g).bfloat16()
    Y0 = torch.randn(B, n, F, generator=g).bfloat16()
    for side, dense in (('fwd', V.t()), ('bwd', V)):
        rp, ci, vals = d[f'{side}_rowptr'], d[f'{side}_colidx'], d[f'{side}_val']
        plan = (d[f'{side}_blk_ptr'], d[f'{side}_blk_cols'], d[f'{side}_blk_vals'])
        for alpha, beta in ((1.0, 0.0), (2.0, -1.0)):
            want = (alpha * torch.einsum('rc,bcf->brf', dense, X.float()) + beta * Y0.float()).bfloat16()
            csr = Y0.clone().cuda()
            hip.csr_spmm_bf16(rp, ci, vals, n, n, cu(X), csr if beta else None, csr, alpha, beta)
            blocked = Y0.clone().cuda()
            hip.csr_spmm_bf16(rp, ci, vals, n, n, cu(X), blocked if beta else None, blocked, alpha, beta, plan=plan)
            assert_one_ulp(csr, want)
            assert_one_ulp(blocked, want)
            assert_one_ulp(blocked, csr)


def test_spmm_bf16_zero_sizes_and_errors(hip):
    from stc_hip._lib import StcError
    rowptr = torch.zeros(5, dtype=torch.int32).cuda()
    colidx = torch.zeros(0, dtype=torch.int32).cuda()
    val = torch.zeros(0).cuda()
    X = torch.ones(1, 4, 64, dtype=torch.bfloat16).cuda()
    Y = torch.full((1, 4, 64), float('nan'), dtype=torch.bfloat16).cuda()
    hip.csr_spmm_bf16(rowptr, colidx, val, 4, 4, X, None, Y, 1.0, 0.0)          # a graph without edges: all zeros
    assert float(Y.float().abs().max()) == 0.0
    with pytest.raises(StcError):                                                # fp32 rows handed to the bf16 entry point
        hip.csr_spmm_bf16(rowptr, colidx, val, 4, 4, X.float(), None, Y, 1.0, 0.0)
    with pytest.raises(StcError):                                                # F not a multiple of 8
        hip.csr_spmm_bf16(rowptr, colidx, val, 4, 4, X[..., :60].contiguous(), None, Y[..., :60].contiguous(), 1.0, 0.0)
    with pytest.raises(StcError):                                                # beta without Y0 (C side)
        hip.csr_spmm_bf16(rowptr, colidx, val, 4, 4, X, None, Y, 1.0, 1.0)


def test_full_size_spmm_bf16_properties(hip):
    """Configuration 5 size (224x224 queen grid, C = 64, L = 32 -> F = 2048 bf16 = the fp32 row's 4 KiB): the row-stochastic
    graph maps a constant field to itself exactly; the adjoint identity ties both orientations; blocked == direct kernel."""
    H = W = 224
    graph = CsrGraph.queen_grid(H, W, normalize=True)
    d = graph.on(torch.device('cuda'))
    N, F = H * W, 2048
    plan_f = (d['fwd_blk_ptr'], d['fwd_blk_cols'], d['fwd_blk_vals'])
    plan_b = (d['bwd_blk_ptr'], d['bwd_blk_cols'], d['bwd_blk_vals'])
    ones = torch.ones(1, N, F, device='cuda', dtype=torch.bfloat16)
    out = torch.empty_like(ones)
    hip.csr_spmm_bf16(d['bwd_rowptr'], d['bwd_colidx'], d['bwd_val'], N, N, ones, None, out, 1.0, 0.0, plan=plan_b)
    assert float((out.float() - 1).abs().max()) == 0.0           # |sum - 1| ~ 1e-7 rounds back to 1 in bf16
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(1, N, F, device='cuda', generator=g).bfloat16()
    y = torch.randn(1, N, F, device='cuda', generator=g).bfloat16()
    STx, STx_direct, Sy = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    hip.csr_spmm_bf16(d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'], N, N, x, None, STx, 1.0, 0.0, plan=plan_f)
    hip.csr_spmm_bf16(d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'], N, N, x, None, STx_direct, 1.0, 0.0)
    assert_one_ulp(STx, STx_direct)
    hip.csr_spmm_bf16(d['bwd_rowptr'], d['bwd_colidx'], d['bwd_val'], N, N, y, None, Sy, 1.0, 0.0, plan=plan_b)
    lhs, rhs = (STx.double() * y.double()).sum(), (x.double() * Sy.double()).sum()
    # each output carries an independent rounding of relative size <= 2^-9: the two sums differ by a random walk over 1e8 terms
    noise = 2.0 ** -9 * float(((STx.double() * y.double()) ** 2).sum() + ((x.double() * Sy.double()) ** 2).sum()) ** 0.5
    assert abs(float(lhs - rhs)) < 8 * noise
    f32 = torch.empty(1, N, F, device='cuda')
    hip.csr_spmm(d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'], N, N, x.float(), None, f32, 1.0, 0.0, plan=plan_f)
    assert_one_ulp(STx, f32.bfloat16())


# ------------------------------------------------------------------ node kernel (projection + category mix) on bf16 slabs
def _node_inputs_bf16(nodes, C, L, Lw, Ho, K, seed):
    g = torch.Generator().manual_seed(seed)
    Zs = [torch.randn(nodes, C, L, generator=g).bfloat16() for _ in range(K)]
    Tc = torch.softmax(torch.randn(K, C, C, generator=g), -1)
    Tc[0] = torch.eye(C)
    for k in range(2, K):
        Tc[k] = 2 * Tc[1] @ Tc[k - 1] - Tc[k - 2]
    W = torch.randn(K * K * Lw, Ho, generator=g) * 0.2
    b = torch.randn(Ho, generator=g)
    return Zs, Tc, W, b


NODE_SHAPES_BF16 = [
    # nodes, C, L, Lw, Ho, K
    (50, 64, 32, 32, 32, 2),      # configuration 5: gates convolution width
    (50, 64, 32, 32, 16, 2),      # candidate width
    (13, 32, 32, 32, 32, 3),
    (21, 32, 16, 16, 16, 1),
    (9, 64, 32, 17, 32, 2),       # padded rows (in + hidden = 17)
    (7, 32, 16, 9, 32, 3),
    (4500, 64, 32, 32, 32, 2),    # persistent grid wraps
    (1, 64, 16, 16, 16, 3),
]


@pytest.mark.parametrize('shape', NODE_SHAPES_BF16)
@pytest.mark.parametrize('bias', [True, False])
def test_bdg_node_fwd_bf16(hip, shape, bias):
    nodes, C, L, Lw, Ho, K = shape
    assert hip.node_bf16_supported(K, K, C, L, Ho) and EM.node_bf16_supported(K, K, C, L, Ho)
    Zs, Tc, W, b = _node_inputs_bf16(nodes, C, L, Lw, Ho, K, seed=nodes + C + L + Ho)
    b = b if bias else None
    want = torch.empty(nodes, C, Ho, dtype=torch.bfloat16)
    EM.bdg_node_fwd_bf16(Zs, Tc, W, b, want)
    got = torch.full((nodes, C, Ho), float('nan'), dtype=torch.bfloat16).cuda()
    hip.bdg_node_fwd_bf16([cu(z) for z in Zs], cu(Tc), cu(W), cu(b), got)
    assert torch.isfinite(got.float()).all()
    # twin with the same rounding points: a flipped rounding of an intermediate U_c moves Y by about one ulp
    assert rel_err(got.float().cpu(), want.float()) < 2.0 ** -7
    assert float((got.float().cpu() - want.float()).abs().mean()) < 2.0 ** -11 * float(want.float().abs().mean()) + 1e-9
    # the exact fp32 math on the same bf16 inputs (emulated fp32 kernel): bf16 rounding of weights / intermediates / output
    exact = torch.empty(nodes, C, Ho)
    EM.bdg_node_fwd([z.float() for z in Zs], Tc, W, b, exact)
    assert rel_err(got.float().cpu(), exact) < 2e-2


@pytest.mark.parametrize('shape', NODE_SHAPES_BF16)
@pytest.mark.parametrize('want_db', [True, False])
def test_bdg_node_bwd_bf16(hip, shape, want_db):
    nodes, C, L, Lw, Ho, K = shape
    Zs, Tc, W, _ = _node_inputs_bf16(nodes, C, L, Lw, Ho, K, seed=3 * nodes + C + L + Ho)
    dY = torch.randn(nodes, C, Ho, generator=torch.Generator().manual_seed(nodes)).bfloat16()
    w_dZ = [torch.empty(nodes, C, L, dtype=torch.bfloat16) for _ in range(K)]
    w_dW, w_db = torch.empty_like(W), torch.empty(Ho)
    EM.bdg_node_bwd_bf16(Zs, Tc, W, dY, w_dZ, w_dW, w_db)
    g_dZ = [torch.full((nodes, C, L), float('nan'), dtype=torch.bfloat16).cuda() for _ in range(K)]
    g_dW = torch.full_like(W, float('nan')).cuda()
    g_db = torch.full((Ho,), float('nan')).cuda() if want_db else None
    hip.bdg_node_bwd_bf16([cu(z) for z in Zs], cu(Tc), cu(W), cu(dY), g_dZ, g_dW, g_db)
    for n in range(K):
        assert torch.isfinite(g_dZ[n].float()).all()
        assert rel_err(g_dZ[n].float().cpu(), w_dZ[n].float()) < 2.0 ** -7
        if Lw < L:
            assert float(g_dZ[n][..., Lw:].float().abs().max()) == 0.0          # pad columns get zero gradient
    assert rel_err(g_dW.cpu(), w_dW) < 2e-3                                     # fp32 sums of bf16-rounded products
    if want_db:
        assert rel_err(g_db.cpu(), w_db) < 1e-5                                 # exact operands, fp32 sums
    # against the exact fp32 math on the same bf16 inputs
    e_dZ = [torch.empty(nodes, C, L) for _ in range(K)]
    e_dW, e_db = torch.empty_like(W), torch.empty(Ho)
    EM.bdg_node_bwd([z.float() for z in Zs], Tc, W, dY.float(), e_dZ, e_dW, e_db, None)
    for n in range(K):
        assert rel_err(g_dZ[n].float().cpu(), e_dZ[n]) < 2e-2
    assert rel_err(g_dW.cpu(), e_dW) < 2e-2
    # bitwise reproducible weight gradients
    g_dW2 = torch.empty_like(g_dW)
    hip.bdg_node_bwd_bf16([cu(z) for z in Zs], cu(Tc), cu(W), cu(dY), g_dZ, g_dW2, g_db)
    assert torch.equal(g_dW, g_dW2)


def test_node_bf16_unsupported_shapes_are_refused(hip):
    from stc_hip._lib import StcError
    assert not hip.node_bf16_supported(2, 2, 16, 32, 32) and not hip.node_bf16_supported(2, 3, 32, 32, 32)
    assert not hip.node_bf16_supported(2, 2, 32, 24, 32) and not hip.node_bf16_supported(4, 4, 32, 32, 32)
    Zs, Tc, W, b = _node_inputs_bf16(5, 16, 32, 32, 32, 2, seed=1)
    with pytest.raises(StcError):
        hip.bdg_node_fwd_bf16([cu(z) for z in Zs], cu(Tc), cu(W), cu(b), torch.empty(5, 16, 32, dtype=torch.bfloat16).cuda())


# ------------------------------------------------------------------ planar STC_Cell kernels on bf16 planes
BTOL = 2e-2          # max-norm relative error against the fp32 twin on the same bf16-valued inputs (bf16 weights, intermediates, outputs)


def _close(got, want, tol=BTOL):
    g, w = got.float().cpu(), want.float()
    assert torch.isfinite(g).all()
    assert rel_err(g, w) < tol, f'rel err {rel_err(g, w)}'
    assert float((g - w).abs().mean()) < 0.25 * tol * float(w.abs().mean()) + 1e-9       # and not just one lucky maximum


@pytest.mark.parametrize('nodes,C,cin', [(50, 32, 16), (13, 64, 16), (4500, 64, 16), (50, 32, 1), (13, 64, 4), (9, 32, 3), (4500, 64, 1)])
def test_planar_cell_kernels_bf16(hip, nodes, C, cin):
    """Gates forward (+ the fused candidate projection), gates backward with its GRU prologue and the post-aggregation
    backward on bf16 planes, wide (16 + 16) and narrow (layer 0: cin + 16) inputs, against the fp32 CPU twins."""
    bf, h, K = torch.bfloat16, 16, 2
    kb = hip.bf16
    assert kb.cell_planar_supported(K, K, C, h)
    Lw = cin + h
    g = torch.Generator().manual_seed(nodes + C + cin)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    rb = lambda *s_: rnd(*s_).to(bf)                       # bf16-valued inputs
    X, SX, H, SH = rb(nodes, C, cin), rb(nodes, C, cin), rb(nodes, C, h), rb(nodes, C, h)
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    Wg, bg = rnd(K * K * Lw, 2 * h) / (4 * Lw) ** 0.5, rnd(2 * h)
    Wc, bc = rnd(K * K * Lw, h) / (4 * Lw) ** 0.5, rnd(h)
    f = lambda t: t.float()
    U_w, R_w, RH_w, A_w, B_w = (torch.empty(nodes, C, h) for _ in range(5))
    EM.cell_gates_fwd_planar(f(X), f(H), f(SX), f(SH), Tc, Wg, bg, U_w, R_w, RH_w)
    nan = lambda *s_: torch.full(s_, float('nan'), dtype=bf).cuda()
    nanf = lambda *s_: torch.full(s_, float('nan')).cuda()
    U, R, RH, A, Bm = (nan(nodes, C, h) for _ in range(5))
    kb.cell_gates_fwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(bg), U, R, RH, post=(cu(Wc), cu(bc), A, Bm))
    _close(U, U_w); _close(R, R_w); _close(RH, RH_w)
    # the candidate's projection sees the ROUNDED R*H plane the kernel stored
    if cin == h:
        EM.node_post_fwd(f(X), Tc, Wc, bc, A_w, B_w, X2=RH.float().cpu())
    else:
        EM.node_post_fwd(RH.float().cpu(), Tc, Wc, bc, A_w, B_w, X2=f(X))
    _close(A, A_w); _close(Bm, B_w)
    U2, R2, RH2 = (nan(nodes, C, h) for _ in range(3))
    kb.cell_gates_fwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(bg), U2, R2, RH2)                # without the second stage
    assert torch.equal(U2, U) and torch.equal(R2, R) and torch.equal(RH2, RH)

    # ---- gates backward (GRU prologue inside)
    dRH, Cand, dHn = rb(nodes, C, h), torch.tanh(rnd(nodes, C, h)).to(bf), rb(nodes, C, h)
    Ub, Rb = U.cpu(), R.cpu()
    wide = cin == h
    dZ_w = [torch.empty(nodes, C, h) if (wide or i >= 2) else None for i in range(4)]
    dW_w, db_w, dH_w = torch.empty_like(Wg), torch.empty(2 * h), torch.empty(nodes, C, h)
    EM.cell_gates_bwd_planar(f(X), f(H), f(SX), f(SH), Tc, Wg, f(dRH), f(Cand), f(Ub), f(Rb), f(dHn), dZ_w, dW_w, db_w, dH_w)
    dZ = [nan(nodes, C, h) if (wide or i >= 2) else None for i in range(4)]
    dW, db, dH = nanf(*Wg.shape), nanf(2 * h), nan(nodes, C, h)
    kb.cell_gates_bwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(dRH), cu(Cand), cu(Ub), cu(Rb), cu(dHn), dZ, dW, db, dH)
    for a, w in zip(dZ, dZ_w):
        if w is not None:
            _close(a, w)
    _close(dH, dH_w)
    assert rel_err(dW.cpu(), dW_w) < BTOL and rel_err(db.cpu(), db_w) < BTOL
    dW2 = nanf(*Wg.shape)
    kb.cell_gates_bwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(dRH), cu(Cand), cu(Ub), cu(Rb), cu(dHn), dZ, dW2, None, dH)
    assert torch.equal(dW2, dW)                                                                                # reproducible, db optional
    # dH = None: the prologue's share of the previous state is added into the H plane's gradient (what the cell graph uses)
    dZf = [nan(nodes, C, h) if (wide or i >= 2) else None for i in range(4)]
    kb.cell_gates_bwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(dRH), cu(Cand), cu(Ub), cu(Rb), cu(dHn), dZf, dW2, None, None)
    _close(dZf[2], dZ_w[2] + dH_w)
    assert torch.equal(dZf[3], dZ[3]) and (not wide or (torch.equal(dZf[0], dZ[0]) and torch.equal(dZf[1], dZ[1])))

    # ---- post-aggregation backward of the candidate convolution
    dA, dB = rb(nodes, C, h), rb(nodes, C, h)
    RHc = RH.cpu()
    dX_w, dX2_w, dWc_w, dbc_w = torch.empty(nodes, C, h), torch.empty(nodes, C, h), torch.empty_like(Wc), torch.empty(h)
    dX, dX2, dWc, dbc = nan(nodes, C, h), nan(nodes, C, h), nanf(*Wc.shape), nanf(h)
    if wide:
        EM.node_post_bwd(f(X), Tc, Wc, f(dA), f(dB), dX_w, dWc_w, dbc_w, X2=f(RHc), dX2=dX2_w)
        kb.node_post_bwd(cu(X), cu(Tc), cu(Wc), cu(dA), cu(dB), dX, dWc, dbc, X2=cu(RHc), dX2=dX2)
        _close(dX2, dX2_w)
    else:
        EM.node_post_bwd(f(RHc), Tc, Wc, f(dA), f(dB), dX_w, dWc_w, dbc_w, X2=f(X))
        kb.node_post_bwd(cu(RHc), cu(Tc), cu(Wc), cu(dA), cu(dB), dX, dWc, dbc, X2=cu(X))
    _close(dX, dX_w)
    assert rel_err(dWc.cpu(), dWc_w) < BTOL and rel_err(dbc.cpu(), dbc_w) < BTOL


@pytest.mark.parametrize('nodes,C,cin', [(50, 32, 16), (13, 64, 16), (4500, 64, 16), (4500, 32, 16), (50, 32, 1), (9, 32, 3), (4500, 32, 4)])
def test_cell_backward_in_one_launch_bf16(hip, nodes, C, cin):
    """stc_cell_bwd_planar_bf16: candidate + gate / blend + gates backward of one planar cell step per node (dY and R*H formed inside and rounded
    to bf16 once, d(R*H) handed over in fp32, dX = both convolutions' shares, the prologue's share folded into dH) against the fp32 CPU twin on
    the bf16-valued planes, and against the two separate bf16 launches; the forward may skip the R*H plane."""
    bf, h, K = torch.bfloat16, 16, 2
    kb = hip.bf16
    assert kb.cell_bwd_planar_supported(C, h, cin) and not kb.cell_bwd_planar_supported(64, h, 1)
    Lw = cin + h
    g = torch.Generator().manual_seed(nodes + C + cin)
    rnd = lambda *s_: torch.randn(*s_, generator=g)
    rb = lambda *s_: rnd(*s_).to(bf)
    X, SX, H, SH = rb(nodes, C, cin), rb(nodes, C, cin), torch.tanh(rnd(nodes, C, h)).to(bf), rb(nodes, C, h)
    Tc = rnd(K, C, C) / C ** 0.5
    Tc[0] = torch.eye(C)
    Wg, Wc = rnd(K * K * Lw, 2 * h) / (4 * Lw) ** 0.5, rnd(K * K * Lw, h) / (4 * Lw) ** 0.5
    U, R, Cand = torch.sigmoid(rnd(nodes, C, h)).to(bf), torch.sigmoid(rnd(nodes, C, h)).to(bf), torch.tanh(rnd(nodes, C, h)).to(bf)
    dHn, dBm = rb(nodes, C, h), rb(nodes, C, h)
    f = lambda t: t.float()
    wide = cin == h
    dZ_w = [torch.empty(nodes, C, h) if (wide or i >= 2) else None for i in range(4)]
    dWg_w, dWc_w, dbg_w, dbc_w = torch.empty_like(Wg), torch.empty_like(Wc), torch.empty(2 * h), torch.empty(h)
    EM.cell_bwd_planar(f(X), f(H), f(SX), f(SH), Tc, Wg, Wc, f(U), f(R), f(Cand), f(dHn), f(dBm), dZ_w, dWg_w, dbg_w, dWc_w, dbc_w)
    nan = lambda *s_: torch.full(s_, float('nan'), dtype=bf).cuda()
    nanf = lambda *s_: torch.full(s_, float('nan')).cuda()
    dZ = [nan(nodes, C, h) if (wide or i >= 2) else None for i in range(4)]
    dWg, dWc, dbg, dbc = nanf(*Wg.shape), nanf(*Wc.shape), nanf(2 * h), nanf(h)
    ops_ = [cu(t) for t in (X, H, SX, SH, Tc, Wg, Wc, U, R, Cand, dHn, dBm)]
    kb.cell_bwd_planar(*ops_, dZ, dWg, dbg, dWc, dbc)
    for a, w in zip(dZ, dZ_w):
        assert (a is None) == (w is None)
        if w is not None:
            _close(a, w)                      # (the twin keeps dY, R*H and d(R*H) in fp32; the kernel rounds the first two to bf16 as the stored planes are)
    for a, w in ((dWg, dWg_w), (dWc, dWc_w), (dbg, dbg_w), (dbc, dbc_w)):
        assert rel_err(a.cpu(), w) < BTOL
    # the two separate launches on the same operands (dY and R*H as rounded planes, dRH rounded on the way)
    dY, RH = (f(dHn) * f(U) * (1 - f(Cand) ** 2)).to(bf), (f(R) * f(H)).to(bf)
    dRH, dXc, dWc2, dZ2, dWg2 = nan(nodes, C, h), nan(nodes, C, h), nanf(*Wc.shape), [nan(nodes, C, h) if (wide or i >= 2) else None for i in range(4)], nanf(*Wg.shape)
    if wide:
        kb.node_post_bwd(cu(X), cu(Tc), cu(Wc), cu(dY), cu(dBm), dXc, dWc2, None, X2=cu(RH), dX2=dRH)
    else:
        kb.node_post_bwd(cu(RH), cu(Tc), cu(Wc), cu(dY), cu(dBm), dRH, dWc2, None, X2=cu(X))
    kb.cell_gates_bwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), dRH, cu(Cand), cu(U), cu(R), cu(dHn), dZ2, dWg2, None, None)
    assert rel_err(dWc, dWc2) < 1e-5 and rel_err(dWg, dWg2) < BTOL            # same operands for dWc; the gates see dRH in fp32 instead of bf16
    _close(dZ[3], dZ2[3].float().cpu())
    # bitwise reproducible, biases optional
    dZ3, dWg3, dWc3 = [nan(nodes, C, h) if (wide or i >= 2) else None for i in range(4)], nanf(*Wg.shape), nanf(*Wc.shape)
    kb.cell_bwd_planar(*ops_, dZ3, dWg3, None, dWc3, None)
    assert torch.equal(dWg3, dWg) and torch.equal(dWc3, dWc) and torch.equal(dZ3[2], dZ[2])
    # forward with the fused candidate projection and no R*H plane: same U, R, A, Bm as with the plane
    bg, bc = rnd(2 * h), rnd(h)
    outs = [[nan(nodes, C, h) for _ in range(5)] for _ in range(2)]
    for o, with_rh in zip(outs, (True, False)):
        kb.cell_gates_fwd_planar(cu(X), cu(H), cu(SX), cu(SH), cu(Tc), cu(Wg), cu(bg), o[0], o[1], o[2] if with_rh else None, post=(cu(Wc), cu(bc), o[3], o[4]))
    for i in (0, 1, 3, 4):
        assert torch.equal(outs[0][i], outs[1][i])
    assert torch.isnan(outs[1][2]).all()


@pytest.mark.parametrize('batch,grid,C,n_add,dual', [(2, (5, 5), 32, 3, True), (1, (4, 7), 64, 5, False), (2, (40, 56), 64, 0, True), (1, (1, 1), 32, 2, False)])
def test_state_aggregations_bf16(hip, batch, grid, C, n_add, dual):
    """stc_spmm_sum_bf16 (gradient of a state from its pieces, optional blend backward), stc_spmm_blend_fwd_bf16 (Y = A + S.Bm
    with the GRU blend) and stc_gru_blend_bwd_bf16, row-blocked and plain CSR, against the fp32 twins on the same inputs."""
    bf, h = torch.bfloat16, 16
    kb = hip.bf16
    graph = CsrGraph.queen_grid(*grid, normalize=True)
    n = graph.n
    g = torch.Generator().manual_seed(n + C + n_add)
    rb = lambda *s_: torch.randn(*s_, generator=g).to(bf)
    f = lambda t: None if t is None else t.float()
    X, X2 = rb(batch, n, C, h), (rb(batch, n, C, h) if dual else None)
    adds = [(rb(batch, n, C, h), 0) for _ in range(n_add)]
    hst = graph._host
    dev = graph.on(torch.device('cuda'))
    U, Cand = torch.rand(batch, n, C, h, generator=g).to(bf), torch.tanh(torch.randn(batch, n, C, h, generator=g)).to(bf)
    for side in ('bwd', 'fwd'):
        csr = tuple(torch.from_numpy(hst[f'{side}_{k}']) for k in ('rowptr', 'colidx', 'val'))
        plan = (dev[f'{side}_blk_ptr'], dev[f'{side}_blk_cols'], dev[f'{side}_blk_vals'])
        Y_w, dY_w = torch.empty(batch, n, C, h), torch.empty(batch, n, C, h)
        EM.spmm_sum(*csr, None, f(X), f(X2), [(f(t), o) for t, o in adds], Y_w, blend=(f(U), f(Cand), dY_w))
        Cand_w, Hn_w = torch.empty(batch, n, C, h), torch.empty(batch, n, C, h)
        A, H = adds[0][0] if adds else X, X
        EM.spmm_blend_fwd(*csr, None, f(X), f(A), f(U), f(H), Cand_w, Hn_w)
        for pl in (plan, None):
            Y, dY = (torch.full((batch, n, C, h), float('nan'), dtype=bf).cuda() for _ in range(2))
            kb.spmm_sum(*(cu(t) for t in csr), pl, cu(X), cu(X2), [(cu(t), o) for t, o in adds], Y, blend=(cu(U), cu(Cand), dY))
            assert_one_ulp(Y, Y_w.to(bf), max_mismatch=0.05)
            _close(dY, dY_w, tol=2.0 ** -7)
            Y1 = torch.full_like(Y, float('nan'))
            kb.spmm_sum(*(cu(t) for t in csr), pl, cu(X), cu(X2), [(cu(t), o) for t, o in adds], Y1)
            assert torch.equal(Y1, Y)
            Cd, Hn = (torch.full((batch, n, C, h), float('nan'), dtype=bf).cuda() for _ in range(2))
            kb.spmm_blend_fwd(*(cu(t) for t in csr), pl, cu(X), cu(A), cu(U), cu(H), Cd, Hn)
            _close(Cd, Cand_w, tol=2.0 ** -7)
            _close(Hn, Hn_w, tol=2.0 ** -7)
    dC = torch.full((batch, n, C, h), float('nan'), dtype=bf).cuda()
    kb.gru_blend_bwd(cu(X), cu(U), None, cu(Cand), dC, None, None)
    assert_one_ulp(dC, (f(X) * f(U) * (1 - f(Cand) ** 2)).to(bf), max_mismatch=0.05)


# ------------------------------------------------------------------ the whole model with bf16 state storage
@pytest.mark.parametrize('C,layers,T,horizon,cin', [(32, 2, 3, 2, 1), (64, 2, 2, 2, 1), (32, 1, 2, 1, 3), (64, 3, 2, 2, 1)])
def test_model_bf16_storage_tracks_fp32(hip, C, layers, T, horizon, cin):
    """STCGNN(storage_dtype=bfloat16) -- every state, gate and gradient plane in bf16, fp32 parameters -- against the same
    model in fp32 (the parity-checked path) on the same inputs and parameters: prediction within bf16 noise of a 2 x (T +
    horizon)-cell recurrence, every parameter gradient pointing the same way (cosine) with the same size."""
    import STC_GNN as M
    from stc_hip import ops
    Hh, Ww, h, K, B = 6, 7, 16, 2, 2
    torch.manual_seed(C + layers + T)
    graph = CsrGraph.queen_grid(Hh, Ww, normalize=True)
    N = Hh * Ww
    kw = dict(num_nodes=N, num_categories=C, Ks=K, Kc=K, input_dim=cin, hidden_dim=h, num_layers=layers, out_horizon=horizon, graph_mode='csr-fixed')
    m32 = M.STCGNN(**kw).cuda()
    m16 = M.STCGNN(**kw, storage_dtype=torch.bfloat16).cuda()
    m16.load_state_dict(m32.state_dict())
    if cin != 1:                                               # keep the prediction scalar per (node, category)
        m32._head = lambda Hs: torch.sigmoid(Hs.float().sum(-1))
        m16._head = lambda Hs: torch.sigmoid(Hs.float().sum(-1))
    Gc = torch.softmax(torch.randn(C, C), -1).cuda()
    X = (torch.rand(B, T, N, C, cin) < 0.3).float().cuda()           # bf16-exact inputs, as the incident indicators are
    Rw = torch.randn(B, horizon, N, C).cuda()
    calls = []
    real = ops.stc_cell_graph
    ops.stc_cell_graph = lambda *a, **k: (calls.append(a[5][0].dtype), real(*a, **k))[1]
    try:
        out = {}
        for name, m in (('f32', m32), ('bf16', m16)):
            m.zero_grad(set_to_none=True)
            pair = M._graphs(graph, Gc, K, K)
            stacked = m._run_cell_graph(pair, X.to(m.storage_dtype))
            assert stacked is not None and stacked.dtype == m.storage_dtype
            y = m._head(stacked).transpose(0, 1)              # bf16 states go through the bf16 head kernels
            (y * Rw).sum().backward()
            out[name] = (y.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    finally:
        ops.stc_cell_graph = real
    assert calls == [torch.float32, torch.bfloat16]
    (y32, g32), (y16, g16) = out['f32'], out['bf16']
    assert torch.isfinite(y16).all()
    assert float((y16 - y32).abs().max()) < 2e-2                      # predictions are in (0, 1)
    assert set(g16) == set(g32)
    for n in g32:
        a, b = g16[n].flatten().double(), g32[n].flatten().double()
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.995, f'd{n}: cosine {cos}'
        assert abs(float(a.norm() / (b.norm() + 1e-30)) - 1) < 5e-2, f'd{n}: norm ratio {float(a.norm() / b.norm())}'


def test_model_bf16_storage_refuses_what_it_cannot_run():
    import STC_GNN as M
    with pytest.raises(ValueError):
        M.STCGNN(30, 32, 2, 2, 1, 16, 2, 2, storage_dtype=torch.bfloat16)                       # learned graphs: fp32 only
    with pytest.raises(ValueError):
        M.STCGNN(30, 32, 2, 2, 1, 16, 2, 2, graph_mode='csr-fixed', storage_dtype=torch.float16)
    m = M.STCGNN(30, 32, 3, 3, 1, 16, 1, 1, graph_mode='csr-fixed', storage_dtype=torch.bfloat16).cuda()     # K = 3: no bf16 cell kernels
    graph = CsrGraph.queen_grid(5, 6, normalize=True)
    with pytest.raises(ValueError):
        m(X_seq=torch.zeros(1, 2, 30, 32).cuda(), As=graph, Ac=torch.eye(32).cuda())


@pytest.mark.parametrize('shape', [(2, 3, 50, 5), (1, 6, 777, 4), (3, 7), (1, 300000)])
def test_output_head_bf16(hip, shape):
    """stc_head_fwd/bwd_bf16 (bf16 state rows of 16, fp32 y / dy / weight gradient) against the fp32 twin on the same values."""
    h = 16
    g = torch.Generator().manual_seed(sum(shape))
    H = torch.randn(*shape, h, generator=g).bfloat16()
    w, b = torch.randn(h, generator=g), torch.randn(1, generator=g)
    y_w = torch.empty(shape)
    EM.head_fwd(H.float(), w, b, y_w)
    y = torch.full(shape, float('nan')).cuda()
    hip.head_fwd(cu(H), cu(w), cu(b), y)
    assert rel_err(y, y_w) < 1e-5
    dy = torch.randn(*shape, generator=g)
    dH_w, dwb_w = torch.empty(*shape, h), torch.empty(h + 1)
    EM.head_bwd(H.float(), w, y_w, dy, dH_w, dwb_w)
    dH, dwb = torch.full((*shape, h), float('nan'), dtype=torch.bfloat16).cuda(), torch.full((h + 1,), float('nan')).cuda()
    hip.head_bwd(cu(H), cu(w), cu(y_w), cu(dy), dH, dwb)
    assert_one_ulp(dH, dH_w.bfloat16(), max_mismatch=0.05)
    assert rel_err(dwb, dwb_w) < 2e-5
    dwb2 = torch.empty_like(dwb)
    hip.head_bwd(cu(H), cu(w), cu(y_w), cu(dy), dH, dwb2)
    assert torch.equal(dwb, dwb2)


def test_adam_trajectory_with_bf16_storage_tracks_fp32(hip):
    """Eight Adam steps (the reference trainer's lr / weight decay, ComboLoss) with bf16 state storage against the fp32 model
    from the same initial parameters and data: the loss trajectories stay together (bf16 rounding of states and gradients is
    noise to Adam, fp32 master weights take the updates)."""
    import STC_GNN as M
    from stc_hip.loss import ComboLoss
    Hh, Ww, C, h, K, B, T, horizon = 6, 7, 32, 16, 2, 4, 4, 2
    torch.manual_seed(3)
    graph = CsrGraph.queen_grid(Hh, Ww, normalize=True)
    N = Hh * Ww
    kw = dict(num_nodes=N, num_categories=C, Ks=K, Kc=K, input_dim=1, hidden_dim=h, num_layers=2, out_horizon=horizon, graph_mode='csr-fixed')
    m32 = M.STCGNN(**kw).cuda()
    m16 = M.STCGNN(**kw, storage_dtype=torch.bfloat16).cuda()
    m16.load_state_dict(m32.state_dict())
    Gc = torch.softmax(torch.randn(C, C), -1).cuda()
    X = (torch.rand(B, T, N, C) < 0.1635).float().cuda()
    Y = (torch.rand(B, horizon, N, C) < 0.1635).float().cuda()
    crit = ComboLoss()
    traj = {}
    for name, m in (('f32', m32), ('bf16', m16)):
        opt = torch.optim.Adam(m.parameters(), lr=2e-3, weight_decay=1e-4)
        losses = []
        for _ in range(8):
            opt.zero_grad(set_to_none=True)
            loss = crit(m(X_seq=X, As=graph, Ac=Gc), Y)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        traj[name] = losses
    assert traj['f32'][-1] < traj['f32'][0]                                     # it trains
    worst = max(abs(a - b) for a, b in zip(traj['f32'], traj['bf16']))
    assert worst < 5e-3, (worst, traj)
    # the parameters after eight steps: same place up to the accumulated rounding noise
    for (n, p), (_, q) in zip(m32.named_parameters(), m16.named_parameters()):
        assert float((p - q).detach().abs().max()) < 5e-3 + 5e-2 * float(p.detach().abs().max()), n


@pytest.mark.parametrize('K,C,L,Ho', [(2, 64, 32, 32), (3, 32, 32, 16), (1, 32, 16, 32), (3, 64, 16, 16)])
def test_bdg_dif_module_on_bf16_features(hip, K, C, L, Ho):
    """The drop-in ``BDG_Dif`` module fed bfloat16 features (fixed CSR graph): forward and every gradient against the same module on
    the same values in fp32 -- SpMM hops, the feature-side Chebyshev recurrence and the node kernels all run in their bf16 forms."""
    import STC_GNN as M
    torch.manual_seed(K * 100 + C + L)
    Hh, Ww, B = 7, 9, 2
    graph = CsrGraph.queen_grid(Hh, Ww, normalize=True)
    N = Hh * Ww
    conv = M.BDG_Dif(K, K, L, Ho).cuda()
    with torch.no_grad():
        conv.b.copy_(torch.randn(Ho))
    Gc = torch.softmax(torch.randn(C, C), -1).cuda()
    X = torch.randn(B, N, C, L).bfloat16().cuda()
    Rw = torch.randn(B, N, C, Ho).cuda()
    res = {}
    for name, x in (('f32', X.float().requires_grad_()), ('bf16', X.clone().requires_grad_())):
        conv.zero_grad(set_to_none=True)
        y = conv(x, graph, Gc)
        assert y.dtype == x.dtype
        (y.float() * Rw).sum().backward()
        res[name] = (y.detach().float(), x.grad.float(), conv.W.grad.clone(), conv.b.grad.clone())
    for a, b, what in zip(res['bf16'], res['f32'], ('Y', 'dX', 'dW', 'db')):
        assert rel_err(a, b) < 3e-2, (what, rel_err(a, b))
    import pytest as _pt
    with _pt.raises(ValueError):                       # shapes off the bf16 kernels are refused, not silently widened
        M.BDG_Dif(K, K, 24, Ho).cuda()(torch.zeros(1, N, C, 24, dtype=torch.bfloat16).cuda(), graph, Gc)


def test_bf16_entry_points_edge_cases(hip):
    """Zero sizes launch nothing (and zero the weight gradients), bad arguments come back as StcError with the C side's text."""
    import ctypes as C_
    from stc_hip._lib import StcError
    kb, bf, h = hip.bf16, torch.bfloat16, 16
    Cc = 32
    z = lambda *s_: torch.zeros(*s_, dtype=bf).cuda()
    Tc = torch.eye(Cc).repeat(2, 1, 1).cuda()
    Wg, Wc = torch.zeros(4 * 32, 32).cuda(), torch.zeros(4 * 32, 16).cuda()
    # nodes = 0: nothing to do, dW / db zeroed by the backward entry points
    e = z(0, Cc, h)
    kb.cell_gates_fwd_planar(e, e, e, e, Tc, Wg, None, z(0, Cc, h), z(0, Cc, h), z(0, Cc, h))
    dW, db = torch.full((4 * 32, 32), float('nan')).cuda(), torch.full((32,), float('nan')).cuda()
    kb.cell_gates_bwd_planar(e, e, e, e, Tc, Wg, e, e, e, e, e, [z(0, Cc, h) for _ in range(4)], dW, db, None)
    assert float(dW.abs().max()) == 0.0 and float(db.abs().max()) == 0.0
    dWc = torch.full((4 * 32, 16), float('nan')).cuda()
    kb.node_post_bwd(e, Tc, Wc, e, e, z(0, Cc, h), dWc, None, X2=z(0, Cc, h), dX2=z(0, Cc, h))
    assert float(dWc.abs().max()) == 0.0
    hip.bdg_node_fwd_bf16([z(0, Cc, 32)], Tc[:1], torch.zeros(32, 16).cuda(), None, z(0, Cc, 16))
    # host-side shape / dtype checks
    X = z(5, Cc, h)
    with pytest.raises(StcError):                                  # fp32 plane handed to the bf16 front
        kb.cell_gates_fwd_planar(X.float(), X, X, X, Tc, Wg, None, z(5, Cc, h), z(5, Cc, h), z(5, Cc, h))
    with pytest.raises(StcError):                                  # input plane of 7 columns
        kb.cell_gates_fwd_planar(z(5, Cc, 7), X, z(5, Cc, 7), X, Tc, torch.zeros(4 * 23, 32).cuda(), None, z(5, Cc, h), z(5, Cc, h), z(5, Cc, h))
    with pytest.raises(StcError):                                  # state copies are an fp32-path feature
        kb.spmm_blend_fwd(None, None, None, None, X, X, X, X, X, X, copies=[(X, 0)])
    # C side: unsupported category count, misaligned plane, null pointer
    lib = hip.lib
    s0 = torch.cuda.current_stream().cuda_stream
    rc = lib.stc_cell_gates_fwd_planar_bf16(X.data_ptr(), X.data_ptr(), X.data_ptr(), X.data_ptr(), Tc.data_ptr(), Wg.data_ptr(), None,
                                            X.data_ptr(), X.data_ptr(), X.data_ptr(), None, None, None, None, 5, 48, 32, 16, s0)
    assert rc == -4 and b'C=48' in lib.stc_last_error()            # STC_EUNSUPPORTED
    big = z(6, Cc, h)
    off = big.view(-1)[4:4 + 5 * Cc * h]                           # 8-byte offset: not 16-byte aligned
    rc = lib.stc_cell_gates_fwd_planar_bf16(X.data_ptr(), off.data_ptr(), X.data_ptr(), X.data_ptr(), Tc.data_ptr(), Wg.data_ptr(), None,
                                            X.data_ptr(), X.data_ptr(), X.data_ptr(), None, None, None, None, 5, Cc, 32, 16, s0)
    assert rc == -2                                                # STC_EALIGN
    rc = lib.stc_spmm_sum_bf16(None, None, None, None, None, None, 4, 4, None, None, 0, None, X.data_ptr(), None, None, None, 1, Cc, 16, s0)
    assert rc == -1                                                # STC_EINVAL: no graph, no operand
    rc = lib.stc_head_fwd_bf16(X.data_ptr(), Tc.data_ptr(), Tc.data_ptr(), Tc.data_ptr(), 5, 8, s0)
    assert rc == -4                                                # hidden width 8: not built
    assert lib.stc_gru_blend_bwd_bf16(X.data_ptr(), X.data_ptr(), X.data_ptr(), X.data_ptr(), 12, s0) == -1       # n not a multiple of 8
    torch.cuda.synchronize()
