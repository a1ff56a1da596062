"""The patch form of the plain spatial aggregation Y = alpha S.X + beta Y0 (stc_patch_spmm_f32, csrc/stc_spmm_patch.hip; reference
STC_GNN.py:37, torch.einsum('bncl,nm->bmcl', X, T_n(Gs))): source rows of a cluster of up to 32 output rows staged through LDS.

CPU: the plan (stc_hip/graph.py _patch_plan) as data -- every row in exactly one patch, source lists within the LDS tile, the plan's
own product (numpy) equal to the CSR product; which graphs get a plan and which do not.
GPU: the kernel against the row-blocked kernel BIT FOR BIT (each row's sum runs over its entries in CSR order with one fmaf each in
both) and against a float64 dense product within 1e-5; ragged patches, empty rows, one and several column chunks, the in-place Y0
epilogue, both orientations of the graph, a renumbered (reverse Cuthill-McKee) node order; what the cell graph launches.
"""
import numpy as np
import pytest
import torch

from stc_hip import CsrGraph
from stc_hip.graph import PATCH_MAX_SRC, PATCH_ROWS, PATCH_WAVES, _patch_plan
from tests.conftest import rel_err

TOL = 1e-5


def _weighted_grid(H, W, seed, permute_seed=None, drop=0.15):
    """H x W 8-neighbour grid with random weights, ~15 % of the edges removed (ragged rows), rows 3 and 10 without entries."""
    g0 = CsrGraph.queen_grid(H, W, normalize=False, permute_seed=permute_seed)
    G = g0.to_dense()
    gen = torch.Generator().manual_seed(seed)
    G = G * torch.randn(G.shape, generator=gen) * (torch.rand(G.shape, generator=gen) > drop)
    G[3] = 0
    G[10] = 0
    return CsrGraph.from_dense(G), G


def _plan_product(h, side, X):
    src, rows, cnt, idx, val = (h[f'{side}_pt_{k}'] for k in ('src', 'rows', 'cnt', 'idx', 'val'))
    Y = np.zeros_like(X)
    for p in range(rows.shape[0]):
        s = src[p].T.reshape(-1)                                                     # position q of the list at [q % waves][q / waves]
        for r in range(PATCH_ROWS):
            if rows[p, r] >= 0:
                Y[rows[p, r]] = (val[p, r, :, None].astype(np.float64) * X[s[idx[p, r]]]).sum(0) if cnt[p, r] else 0.0
    return Y


@pytest.mark.parametrize('H,W,permute', [(30, 30, None), (17, 41, None), (40, 40, 7)])
def test_patch_plan_is_the_matrix(H, W, permute):
    graph, G = _weighted_grid(H, W, seed=H * W, permute_seed=permute)
    if permute is not None:
        graph, order = graph.with_locality()
        assert order is not None
        G = G[order][:, order]
    h, n = graph._host, graph.n
    X = np.random.default_rng(1).standard_normal((n, 5))
    for side, dense in (('fwd', G.t()), ('bwd', G)):
        src, rows, nsrc = h[f'{side}_pt_src'], h[f'{side}_pt_rows'], h[f'{side}_pt_nsrc']
        first = np.full(n, -1)                                                      # the patch each row is in: exactly one
        for p in range(rows.shape[0]):
            mine = np.unique(rows[p][rows[p] >= 0])
            assert (first[mine] == -1).all()
            first[mine] = p
            for r in range(PATCH_ROWS):                                             # a repeated slot: the same wave's first row, tables and all
                if rows[p, r] >= 0 and r >= mine.size:
                    f = r % PATCH_WAVES
                    assert rows[p, r] == rows[p, f] and np.array_equal(h[f'{side}_pt_idx'][p, r], h[f'{side}_pt_idx'][p, f])
                    assert np.array_equal(h[f'{side}_pt_val'][p, r], h[f'{side}_pt_val'][p, f])
        assert (first >= 0).all()
        assert src.shape[1:] == (PATCH_WAVES, PATCH_MAX_SRC // PATCH_WAVES) and nsrc.max() <= PATCH_MAX_SRC and nsrc.min() >= 0
        for p in range(rows.shape[0]):                                              # distinct source rows, then repeats of the first
            lst = src[p].T.reshape(-1)
            assert np.unique(lst[:nsrc[p]]).size == nsrc[p] and (lst[nsrc[p]:] == lst[0]).all()
        assert h[f'{side}_pt_idx'].dtype == np.uint8 and h[f'{side}_pt_idx'].shape[2] % 4 == 0
        np.testing.assert_allclose(_plan_product(h, side, X), dense.double().numpy() @ X, rtol=0, atol=1e-12)
    fetch, per_patch = graph.patch_stats['fwd']
    assert fetch < 2.4 and per_patch > 20


def test_which_graphs_get_a_patch_plan():
    assert CsrGraph.queen_grid(224, 224).patch_stats['fwd'][0] < 2.0                 # the bench's graph: 1.98 source rows per output row
    assert CsrGraph.queen_grid(40, 40, permute_seed=1).patch_stats == {}             # clusters exist, but their rows lie all over the plane
    assert CsrGraph.queen_grid(40, 40, permute_seed=1).with_locality()[0].patch_stats['fwd'][0] < 2.2      # ... renumbered: runs of rows
    g = torch.Generator().manual_seed(0)
    rows, cols = torch.randint(0, 2000, (2, 16000), generator=g).numpy()
    keep = np.unique(rows.astype(np.int64) * 2000 + cols, return_index=True)[1]
    assert CsrGraph(2000, rows[keep], cols[keep], np.ones(keep.size)).patch_stats == {}      # a random graph: patches of a few rows
    dense = torch.rand(80, 80, generator=g)
    assert CsrGraph.from_dense(dense).patch_stats == {}                               # rows of 80 entries: wider than the table
    # a k-nearest-neighbour graph: out-lists of exactly 8, in-lists of 8 on average but up to ~20 -> only the orientation with even rows gets a plan
    from scipy.spatial import cKDTree
    pts = np.random.default_rng(0).random((4000, 2))
    nb = cKDTree(pts).query(pts, k=9)[1][:, 1:]
    knn = CsrGraph(4000, np.repeat(np.arange(4000), 8), nb.ravel(), np.ones(32000)).with_locality()[0]
    assert knn.patch_stats == {}                                                     # (in-lists: skewed lengths; out-lists: scattered rows)
    rp = np.array([0, 0, 0], dtype=np.int32)
    assert _patch_plan(rp, np.zeros(0, np.int32), np.zeros(0, np.float32), 2) is None


def test_graphs_build_without_scipy(monkeypatch):
    """scipy is optional in every plan: a lattice with self-loops (9 entries per row: tiles are found without scipy, the two-ring tables of
    width 8 do not fit, and the ring-bounded clusters would want scipy) still builds, with the two separate launches left in place."""
    import sys
    for name in [m for m in sys.modules if m == 'scipy' or m.startswith('scipy.')]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.setitem(sys.modules, 'scipy', None)
    monkeypatch.setitem(sys.modules, 'scipy.sparse', None)
    with pytest.raises(ImportError):
        import scipy.sparse  # noqa: F401
    g0 = CsrGraph.queen_grid(40, 40, normalize=False)
    h = g0._host
    rows = np.repeat(np.arange(1600), np.diff(h['bwd_rowptr'].astype(np.int64)))
    r, c = np.append(rows, np.arange(1600)), np.append(h['bwd_colidx'], np.arange(1600))
    g = CsrGraph(1600, r, c, np.ones(r.size, dtype=np.float32))
    assert g.nnz == g0.nnz + 1600 and 'fwd_pt_rows' in g._host
    assert not any(k.startswith('fwd_r2') or k.startswith('bwd_r2') for k in g._host) and not g.ring2_clusters


def test_grid_graphs_get_tile_patches():
    """A graph that is a 4- / 8-neighbour lattice in its node numbering gets 4 x 8 tiles (every slot of an interior patch used, 60 source rows);
    one long-range edge, or a random numbering, and the greedy clusters take over."""
    from stc_hip.graph import _grid_tiles
    g = CsrGraph.queen_grid(224, 224)
    assert g.patch_stats['fwd'] == (pytest.approx(1.8505, abs=1e-3), 32.0) and g._host['fwd_pt_rows'].shape[0] == 56 * 28
    assert (g._host['fwd_pt_nsrc'].max(), g._host['fwd_pt_rows'].min()) == (60, 0)
    h = CsrGraph.queen_grid(64, 48)._host
    assert _grid_tiles(h['fwd_rowptr'].astype(np.int64), h['fwd_colidx'], 64 * 48) is not None
    rook = [(i, j) for i in range(40 * 30) for j in (i - 30, i - 1, i + 1, i + 30) if 0 <= j < 1200 and (abs(j - i) == 30 or j // 30 == i // 30)]
    r, c = np.array(rook).T
    assert CsrGraph(1200, r, c, np.ones(r.size)).patch_stats['fwd'][1] > 28                       # 4-neighbour lattice: tiles as well
    p = CsrGraph.queen_grid(40, 40, permute_seed=3)._host
    assert _grid_tiles(p['fwd_rowptr'].astype(np.int64), p['fwd_colidx'], 1600) is None
    band = [(i, j) for i in range(1001) for j in range(max(0, i - 2), min(1001, i + 3)) if i not in (3, 4, 5, 6, 7, 8, 9, 10)]      # a lattice 2 wide, rows 3..10 empty
    br, bc = np.array(band).T
    assert CsrGraph(1001, br, bc, np.ones(br.size)).patch_stats['fwd'][1] > 16                    # ... too narrow for tiles: clusters
    far = CsrGraph(1200, np.append(r, 5), np.append(c, 900), np.ones(r.size + 1))._host          # one edge across the lattice
    assert _grid_tiles(far['fwd_rowptr'].astype(np.int64), far['fwd_colidx'], 1200) is None


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,F,B,permute', [(30, 30, 512, 2, None), (17, 41, 256, 3, None), (40, 40, 1024, 1, 7), (12, 12, 768, 2, None), (224, 8, 512, 1, None)])
def test_patch_spmm_equals_the_row_blocked_kernel(H, W, F, B, permute):
    from stc_hip._lib import HipKernels, KernelTimer
    hip = HipKernels()
    hip.patch_min_items = 0                                                        # (small launches go to the row-blocked kernel by default)
    graph, G = _weighted_grid(H, W, seed=H + W + F, permute_seed=permute)
    if permute is not None:
        graph, order = graph.with_locality()
        G = G[order][:, order]
    n = graph.n
    d = graph.on(torch.device('cuda'))
    gen = torch.Generator().manual_seed(F)
    X = torch.randn(B, n, F, generator=gen)
    Y0 = torch.randn(B, n, F, generator=gen)
    for side, dense in (('fwd', G.t()), ('bwd', G)):
        rp, ci, vals = d[f'{side}_rowptr'], d[f'{side}_colidx'], d[f'{side}_val']
        blocks = (d[f'{side}_blk_ptr'], d[f'{side}_blk_cols'], d[f'{side}_blk_vals'])
        patches = blocks + (tuple(d[f'{side}_pt_{k}'] for k in ('src', 'rows', 'cnt', 'idx', 'val')),)
        for alpha, beta in ((1.0, 0.0), (2.0, -1.0), (1.0, 1.0)):
            want = alpha * torch.einsum('rc,bcf->brf', dense.double(), X.double()) + beta * Y0.double()
            blocked = Y0.clone().cuda()
            hip.csr_spmm(rp, ci, vals, n, n, X.cuda(), blocked if beta else None, blocked, alpha, beta, plan=blocks)
            patched = Y0.clone().cuda()
            hip.timer = t = KernelTimer()
            hip.csr_spmm(rp, ci, vals, n, n, X.cuda(), patched if beta else None, patched, alpha, beta, plan=patches)      # (in place when there is a Y0)
            hip.timer = None
            assert list(t.summary()) == ['stc_patch_spmm_f32']
            assert rel_err(patched, want) < TOL
            assert torch.equal(patched, blocked)


@pytest.mark.gpu
def test_patch_spmm_rejects_what_it_cannot_do():
    from stc_hip._lib import HipKernels, KernelTimer, StcError
    hip = HipKernels()
    hip.patch_min_items = 0
    graph, _ = _weighted_grid(12, 12, seed=5)
    d = graph.on(torch.device('cuda'))
    pt = tuple(d[f'fwd_pt_{k}'] for k in ('src', 'rows', 'cnt', 'idx', 'val'))
    n = graph.n
    X, Y = torch.randn(1, n, 320).cuda(), torch.empty(1, n, 320).cuda()
    plan = (d['fwd_blk_ptr'], d['fwd_blk_cols'], d['fwd_blk_vals'], pt)
    hip.timer = t = KernelTimer()
    hip.csr_spmm(d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'], n, n, X, None, Y, 1.0, 0.0, plan=plan)      # 320 floats: not whole chunks -> row-blocked
    hip.timer = None
    assert list(t.summary()) == ['stc_bcsr_spmm_f32']
    import ctypes
    p = lambda a: ctypes.c_void_p(a.data_ptr())
    rc = hip.lib.stc_patch_spmm_f32(*[p(a) for a in pt], pt[3].shape[0], pt[3].shape[2], n, n, p(X), None, p(Y), 1, 320, 1.0, 0.0, None)
    assert rc != 0 and b'multiple of 256' in hip.lib.stc_last_error()
    with pytest.raises(StcError):
        hip.csr_spmm(d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'], n, n, torch.randn(1, n, 256).cuda(), None, torch.empty(1, n, 256).cuda(), 1.0, 0.0,
                     plan=plan[:3] + ((pt[0], pt[1], pt[2], pt[3].to(torch.int32), pt[4]),))


@pytest.mark.gpu
def test_small_launches_keep_the_row_blocked_kernel():
    """Default dispatch: the patch form from 3 000 (patch, sample) workgroups on (1 000 with a Y0 operand), the row-blocked one below."""
    from stc_hip._lib import HipKernels, KernelTimer
    hip = HipKernels()
    graph = CsrGraph.queen_grid(64, 64)
    d = graph.on(torch.device('cuda'))
    n, n_p = graph.n, d['fwd_pt_idx'].shape[0]
    plan = (d['fwd_blk_ptr'], d['fwd_blk_cols'], d['fwd_blk_vals'], tuple(d[f'fwd_pt_{k}'] for k in ('src', 'rows', 'cnt', 'idx', 'val')))
    for B, want in ((1, 'stc_bcsr_spmm_f32'), (-(-3000 // n_p), 'stc_patch_spmm_f32')):
        X, Y = torch.randn(B, n, 256).cuda(), torch.empty(B, n, 256).cuda()
        hip.timer = t = KernelTimer()
        hip.csr_spmm(d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'], n, n, X, None, Y, 1.0, 0.0, plan=plan)
        hip.timer = None
        assert list(t.summary()) == [want]


@pytest.mark.gpu
def test_patch_spmm_at_the_bench_size():
    """BASELINE.json's metric shape (224 x 224 grid, 5 samples, rows of 512 floats; 7 970 workgroups): the default dispatch takes the patch
    kernel, and its result equals the row-blocked kernel's bit for bit in both orientations; a checksum of checksums against float64 on a
    sample of rows (the dense product does not fit)."""
    from stc_hip._lib import HipKernels, KernelTimer
    hip = HipKernels()
    graph = CsrGraph.queen_grid(224, 224)
    d = graph.on(torch.device('cuda'))
    n, B, F = graph.n, 5, 512
    gen = torch.Generator().manual_seed(11)
    X = torch.randn(B, n, F, generator=gen).cuda()
    for side in ('fwd', 'bwd'):
        rp, ci, vals = d[f'{side}_rowptr'], d[f'{side}_colidx'], d[f'{side}_val']
        blocks = (d[f'{side}_blk_ptr'], d[f'{side}_blk_cols'], d[f'{side}_blk_vals'])
        patches = blocks + (tuple(d[f'{side}_pt_{k}'] for k in ('src', 'rows', 'cnt', 'idx', 'val')),)
        Yb, Yp = torch.empty_like(X), torch.empty_like(X)
        hip.csr_spmm(rp, ci, vals, n, n, X, None, Yb, 1.0, 0.0, plan=blocks)
        hip.timer = t = KernelTimer()
        hip.csr_spmm(rp, ci, vals, n, n, X, None, Yp, 1.0, 0.0, plan=patches)
        hip.timer = None
        assert list(t.summary()) == ['stc_patch_spmm_f32']
        assert torch.equal(Yp, Yb)
        rows = torch.randint(0, n, (64,), generator=gen)
        rpc, cic, vc = rp.cpu().long(), ci.cpu().long(), vals.cpu().double()
        for r in rows.tolist():
            cols = cic[rpc[r]:rpc[r + 1]]
            want = (vc[rpc[r]:rpc[r + 1], None, None] * X[:, cols].cpu().double().transpose(0, 1)).sum(0)      # (B, F)
            assert rel_err(Yp[:, r].cpu(), want) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,F,B,permute', [(30, 30, 1024, 2, None), (17, 41, 512, 3, None), (40, 40, 2048, 1, 7)])
def test_patch_spmm_bf16_equals_the_row_blocked_kernel(H, W, F, B, permute):
    """bf16 rows (BASELINE configuration 5's storage): stc_patch_spmm_bf16 against stc_bcsr_spmm_bf16 bit for bit (fp32 sums in CSR
    order, one rounding), and against float64 on the bf16-valued inputs to a bf16 ulp."""
    from stc_hip._lib import HipKernels, KernelTimer
    hip = HipKernels()
    hip.patch_min_items = 0
    graph, G = _weighted_grid(H, W, seed=H + W + F, permute_seed=permute)
    if permute is not None:
        graph, order = graph.with_locality()
        G = G[order][:, order]
    n = graph.n
    d = graph.on(torch.device('cuda'))
    gen = torch.Generator().manual_seed(F)
    X = torch.randn(B, n, F, generator=gen).to(torch.bfloat16)
    Y0 = torch.randn(B, n, F, generator=gen).to(torch.bfloat16)
    for side, dense in (('fwd', G.t()), ('bwd', G)):
        rp, ci, vals = d[f'{side}_rowptr'], d[f'{side}_colidx'], d[f'{side}_val']
        blocks = (d[f'{side}_blk_ptr'], d[f'{side}_blk_cols'], d[f'{side}_blk_vals'])
        patches = blocks + (tuple(d[f'{side}_pt_{k}'] for k in ('src', 'rows', 'cnt', 'idx', 'val')),)
        for alpha, beta in ((1.0, 0.0), (2.0, -1.0)):
            want = alpha * torch.einsum('rc,bcf->brf', dense.double(), X.double()) + beta * Y0.double()
            blocked = Y0.clone().cuda()
            hip.csr_spmm(rp, ci, vals, n, n, X.cuda(), blocked if beta else None, blocked, alpha, beta, plan=blocks)
            patched = Y0.clone().cuda()
            hip.timer = t = KernelTimer()
            hip.csr_spmm(rp, ci, vals, n, n, X.cuda(), patched if beta else None, patched, alpha, beta, plan=patches)
            hip.timer = None
            assert list(t.summary()) == ['stc_patch_spmm_bf16']
            assert torch.equal(patched, blocked)
            assert rel_err(patched.float(), want) < 2.0 ** -8                           # one bf16 rounding of the result
