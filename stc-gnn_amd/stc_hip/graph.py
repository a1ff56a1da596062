"""Spatial-graph containers for the HIP path.

The reference only ever multiplies by a dense, learned ``Gs`` (STC_GNN.py:37).
Here the spatial operand is always a CSR pair with int32 indices in HBM:

* ``CsrGraph``        a fixed sparse graph supplied by the caller (``csr-fixed`` mode), holding
                      CSR(Gs^T) for the forward 1-mode product ``Gs^T . X`` and CSR(Gs) for its
                      backward (the reference's einsum is the TRANSPOSED aggregation);
* ``dense_operand``   the learned dense ``Gs`` of the reference viewed as a CSR matrix with the
                      full N x N pattern (``dense-learned`` mode, small N), values differentiable.

Index arrays are validated on the host when a graph is built: the kernels trust them.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

INT32_MAX = 2 ** 31 - 1


def _csr_from_coo(rows: np.ndarray, cols: np.ndarray, vals: np.ndarray, n: int):
    """Sort COO by (row, col) and build rowptr; returns (rowptr, colidx, vals, order)."""
    order = np.lexsort((cols, rows))
    r, c = rows[order], cols[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, r + 1, 1)
    np.cumsum(rowptr, out=rowptr)
    return rowptr.astype(np.int32), c.astype(np.int32), vals[order].astype(np.float32), order


BLOCK_ROWS = 4      # = STC_SPMM_BLOCK_ROWS of include/stc_hip.h
BLOCK_BATCH = 6     # = PU of csrc/stc_spmm.hip: neighbour rows per pipelined gather batch


def _row_block_plan(rowptr: np.ndarray, colidx: np.ndarray, val: np.ndarray, n: int):
    """Row-blocked (BCSR, BLOCK_ROWS x 1) form of a CSR matrix for ``stc_bcsr_spmm_f32``.

    Per block of BLOCK_ROWS consecutive rows: the sorted distinct columns its rows touch (blk_ptr, blk_cols)
    and, per column, one value per row of the block (zero where that row has no such entry).  A neighbour
    row shared by several rows of a block is then fetched once per block instead of once per row.
    """
    n_blocks = (n + BLOCK_ROWS - 1) // BLOCK_ROWS
    row_of = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr.astype(np.int64)))
    blk_of = row_of // BLOCK_ROWS
    key = blk_of * max(n, 1) + colidx.astype(np.int64)
    uniq, inverse = np.unique(key, return_inverse=True)           # sorted by block, then column
    blk_ptr = np.searchsorted(uniq // max(n, 1), np.arange(n_blocks + 1))
    vals = np.zeros((uniq.size, BLOCK_ROWS), dtype=np.float32)
    vals[inverse, row_of % BLOCK_ROWS] = val
    cols = uniq % max(n, 1)
    distinct = int(uniq.size)
    # Every block's list is padded to a multiple of BLOCK_BATCH entries with zero-weight repeats of its last column (a row the
    # block fetches anyway: the repeat is an L1 / L2 hit): the kernel's pipelined gather then runs in whole batches with no
    # remainder loop.  The 8-neighbour grid's interior blocks list 18 columns and need no padding at all.
    counts = np.diff(blk_ptr)
    padded = -(-counts // BLOCK_BATCH) * BLOCK_BATCH
    if (padded != counts).any():
        new_ptr = np.concatenate([[0], np.cumsum(padded)])
        dst = np.repeat(new_ptr[:-1] - blk_ptr[:-1], counts) + np.arange(uniq.size)      # where each real entry goes
        out_cols = np.zeros(int(new_ptr[-1]), dtype=np.int64)
        out_vals = np.zeros((int(new_ptr[-1]), BLOCK_ROWS), dtype=np.float32)
        out_cols[dst] = cols
        out_vals[dst] = vals
        pad_n = padded - counts
        has = counts > 0
        pad_dst = np.repeat(new_ptr[:-1] + counts, pad_n) + (np.arange(int(pad_n.sum())) - np.repeat(np.cumsum(pad_n) - pad_n, pad_n))
        last_col = np.where(has, cols[np.maximum(blk_ptr[1:] - 1, 0)] if uniq.size else 0, 0)
        out_cols[pad_dst] = np.repeat(last_col, pad_n)
        cols, vals, blk_ptr = out_cols, out_vals, new_ptr
    return dict(blk_ptr=blk_ptr.astype(np.int32), blk_cols=cols.astype(np.int32), blk_vals=vals, distinct=distinct)


PATCH_ROWS = 32         # = STC_PATCH_ROWS of include/stc_hip.h
PATCH_MAX_SRC = 64      # = STC_PATCH_MAX_SRC
PATCH_MAX_WIDTH = 32    # = STC_PATCH_MAX_WIDTH
PATCH_WAVES = 4         # waves of the kernel's workgroup: slot r of a patch is wave r % 4's
PATCH_WIDTHS = (4, 8, 12, 16, 24, 32)      # entries per row the kernel is built for


def _patch_plan(rowptr: np.ndarray, colidx: np.ndarray, val: np.ndarray, n: int, max_fetch: float = 2.6, min_rows: float = 16.0):
    """Patch form of a CSR matrix for ``stc_patch_spmm_f32`` (include/stc_hip.h), or None when the graph does not cluster.

    A graph that is a lattice in its node numbering gets 4 x 8 tiles (``_grid_tiles``).  Otherwise rows are grouped greedily: the first row not yet in a patch seeds one, which grows breadth-first over the symmetrised pattern
    (rows not yet taken only) until it holds PATCH_ROWS rows or the distinct columns its rows touch would pass PATCH_MAX_SRC.  On a
    mesh the patches come out as compact blobs (the 8-neighbour grid: 31.5 rows and 62 source rows per patch, 1.98 source rows per
    output row; ideal 4 x 8 tiles would have 1.88).  A graph without such locality -- patches of a few rows, more than ``max_fetch``
    source rows per output row -- or with a row of more than PATCH_MAX_WIDTH entries gets no plan: the row-blocked kernel is the
    better one there.  Deterministic in the arrays alone (every rank of a job builds the same plan).

    The patches keep the order they were seeded in.  The kernel's workgroups are dispatched in that order, an eighth of the list per XCD,
    and what neighbouring patches share is found in that XCD's L2 when they are close in the list (measured on the bench's grid, whose
    patches are seeded row by row: FETCH_SIZE 1.14 x the matrix).  Re-ordering the list as eight narrow sweeps (reverse Cuthill-McKee of
    the patch graph, cut in eight, each part ordered again) was tried: no faster on a grid in natural or renumbered order, three times
    slower on a graph numbered at random (tools/probes/patch_spmm_unit.py).
    """
    if n == 0 or colidx.size == 0:
        return None
    rp = rowptr.astype(np.int64)
    deg = np.diff(rp)
    if int(deg.max()) > PATCH_MAX_WIDTH:
        return None
    # the kernel sums a FIXED number of entries per row (the table width, zero-weight padding included): a few long rows among short ones -- the
    # in-neighbour lists of a k-nearest-neighbour graph: 8 on average, up to ~20 -- would make every row pay for the longest (measured: 793
    # against 264 us row-blocked).  Such a matrix keeps the row-blocked form.
    if next(w for w in PATCH_WIDTHS if w >= int(deg.max())) > max(8.0, 1.5 * float(deg.mean())):
        return None
    rpl, cil = rp.tolist(), colidx.tolist()
    tiles = _grid_tiles(rp, colidx, n)
    if tiles is not None and n / len(tiles) >= min_rows and sum(len(src) for _, src in tiles) / n <= max_fetch:      # (a narrow lattice: clusters)
        return _patch_tables(rpl, cil, val, n, tiles, int(deg.max()))
    import scipy.sparse as sp
    A = sp.csr_matrix((np.ones(colidx.size, dtype=np.int8), colidx, rp), shape=(n, n))
    S = (A + A.T).tocsr()
    S.sort_indices()                                   # (neighbours in index order whatever the scipy build: every rank grows the same patches)
    srp, sci = S.indptr.tolist(), S.indices.tolist()
    taken = np.zeros(n, dtype=bool)
    patches = []                                       # (rows, source rows in first-touch order)
    for seed in range(n):
        if taken[seed]:
            continue
        rows, src, queue, queued, qi = [], {}, [seed], {seed}, 0
        while qi < len(queue) and len(rows) < PATCH_ROWS:
            u = queue[qi]
            qi += 1
            new = [c for c in cil[rpl[u]:rpl[u + 1]] if c not in src]
            if len(src) + len(new) > PATCH_MAX_SRC:
                continue                               # (rows stay available to later patches; a single row fits: width <= 32)
            for c in new:
                src[c] = len(src)
            rows.append(u)
            taken[u] = True
            for w in sci[srp[u]:srp[u + 1]]:
                if not taken[w] and w not in queued:
                    queued.add(w)
                    queue.append(w)
        patches.append((rows, src))
    if n / len(patches) < min_rows or sum(len(src) for _, src in patches) / n > max_fetch:
        return None
    # ... and whose patches gather RUNS of consecutive rows: the kernel asks HBM for 1 KiB of a row at a time, which it serves at full rate only
    # when neighbouring requests fall into the same DRAM pages.  Mean run length of the sorted source lists -- the bench's grid 9.9 (tiles), the
    # same grid renumbered by reverse Cuthill-McKee 7.6 (197 us against 213 - 225 row-blocked); a k-nearest-neighbour mesh of random points,
    # renumbered: 3.4, and 740 us against 252 row-blocked.  Below 5: no plan.
    n_src = n_runs = 0
    for _, src in patches:
        l = np.sort(np.fromiter(src, dtype=np.int64, count=len(src)))
        n_src += l.size
        n_runs += int((np.diff(l) != 1).sum()) + 1
    if n_src < 5 * n_runs:
        return None
    return _patch_tables(rpl, cil, val, n, patches, int(deg.max()))


GRID_TILE = (4, 8)      # rows x columns of a patch on a graph that is a grid in its node numbering: 32 nodes, 6 x 10 = 60 source rows


def _grid_tiles(rp: np.ndarray, colidx: np.ndarray, n: int):
    """Patches as 4 x 8 tiles when the graph IS a grid in its node numbering -- node i = (i // W, i % W) and every entry joins nodes at most one
    row and one column apart (the 4- / 8-neighbour lattices of city cells the reference's datasets are; ``queen_grid``) -- else None.
    Such tiles are what the greedy clusters approximate: all 32 rows used, 1.88 source rows per output row, and in row-major tile order a tile's
    neighbours above are W / 8 patches back in the list, inside one XCD's L2 (191 against 199 us for the clusters on the bench's unit)."""
    if colidx.size == 0 or n < 64:
        return None
    row_of = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    col = colidx.astype(np.int64)
    reach = int(np.abs(col - row_of).max())
    for W in (reach - 1, reach):                         # 8-neighbour lattice: reach = W + 1; 4-neighbour: reach = W
        if W < 2 or W >= n:
            continue
        dh, dw = np.abs(col // W - row_of // W), np.abs(col % W - row_of % W)
        if int(dh.max()) <= 1 and int(dw.max()) <= 1:
            break
    else:
        return None
    H, (TY, TX) = -(-n // W), GRID_TILE
    rpl, cil = rp.tolist(), colidx.tolist()
    # Tile order = the order the kernel's workgroups are dispatched in, an eighth of the list per XCD (stc_xcd_tile).  The tile grid is cut into 8
    # regions, one per XCD, each listed row by row: a tile's neighbours above are then one region WIDTH back in the list -- 7 tiles (0.85 MB of
    # source rows) on the bench's grid, inside the XCD's 4 MiB L2, where whole rows of 28 tiles (3.4 MB, two chunks in flight) lost part of
    # them: FETCH_SIZE 564 -> 547 MB, 190.7 -> 181.0 us.  The regions have to coincide with the XCDs' shares: strips 4 - 6 or 8 - 10 tiles wide,
    # which straddle them, measured 190 - 198 us.
    TR, TC = -(-H // TY), -(-W // TX)
    gx = min((g for g in (1, 2, 4, 8) if TC >= g and TR >= 8 // g), key=lambda g: abs(TC / g - 7.0), default=1)
    col_parts = np.array_split(np.arange(TC), gx)
    row_parts = np.array_split(np.arange(TR), 8 // gx if TR >= 8 // gx else 1)
    tiles = []
    for cols in col_parts:
        for rws in row_parts:
            for tr in rws.tolist():
                for tc in cols.tolist():
                    ty, tx = tr * TY, tc * TX
                    rows = [y * W + x for y in range(ty, min(ty + TY, H)) for x in range(tx, min(tx + TX, W)) if y * W + x < n]
                    if not rows:
                        continue
                    src = {}
                    for c in sorted({c for u in rows for c in cil[rpl[u]:rpl[u + 1]]}):
                        src[c] = len(src)
                    tiles.append((rows, src))
    return tiles


def _patch_tables(rpl, cil, val, n, patches, max_deg):
    """The arrays of ``stc_patch_spmm_f32`` for a given grouping: ``patches`` = [(rows, {source row: position in the patch's list})]."""
    n_p = len(patches)
    n_src = sum(len(src) for _, src in patches)
    width = next(w for w in PATCH_WIDTHS if w >= max_deg)
    pt_src = np.zeros((n_p, PATCH_WAVES, PATCH_MAX_SRC // PATCH_WAVES), dtype=np.int32)
    pt_nsrc = np.zeros(n_p, dtype=np.int32)
    pt_rows = np.full((n_p, PATCH_ROWS), -1, dtype=np.int32)
    pt_cnt = np.zeros((n_p, PATCH_ROWS), dtype=np.int32)
    pt_idx = np.zeros((n_p, PATCH_ROWS, width), dtype=np.uint8)
    pt_val = np.zeros((n_p, PATCH_ROWS, width), dtype=np.float32)
    for p, (rows, src) in enumerate(patches):
        # position q of the list at [q % 4][q / 4]: the wave that stages it reads its 16 numbers with one scalar load; the record is
        # filled up with repeats of the first source row (staged like the others: no branch in the kernel, a cache hit)
        lst = np.full(PATCH_MAX_SRC, next(iter(src), 0), dtype=np.int32)      # (a patch of rows without entries gathers nothing: any row)
        lst[:len(src)] = list(src)                      # (dicts keep insertion order: position = value)
        pt_src[p] = lst.reshape(PATCH_MAX_SRC // PATCH_WAVES, PATCH_WAVES).T
        pt_nsrc[p] = len(src)
        pt_rows[p, :len(rows)] = rows
        for r, u in enumerate(rows):
            a, b = rpl[u], rpl[u + 1]
            k = b - a
            pt_cnt[p, r] = k
            if k:
                idx = [src[c] for c in cil[a:b]]
                pt_idx[p, r, :k] = idx
                pt_val[p, r, :k] = val[a:b]
                pt_idx[p, r, k:] = idx[-1]              # the tail: zero-weight repeats of the last entry (a row it sums anyway)
    # Slot r of a patch belongs to wave r % 4 of the workgroup (its i-th row: r = wave + 4 i).  A wave's slots past the patch's rows
    # repeat that wave's FIRST row, tables and all: the kernel's row loop then needs no "is there a row" branch (a wave computes and
    # stores the same values twice, in program order; with Y0 aliasing Y both read Y0 before either stores).  Only a wave with no row at
    # all (patches of 1 .. 3 rows) keeps -1 slots, which the kernel sends to a dump line.
    n_in = (pt_rows >= 0).sum(1)
    for p in np.nonzero(n_in < PATCH_ROWS)[0]:
        for r in range(int(n_in[p]), PATCH_ROWS):
            if r % PATCH_WAVES < n_in[p]:
                first = r % PATCH_WAVES
                pt_rows[p, r], pt_cnt[p, r], pt_idx[p, r], pt_val[p, r] = pt_rows[p, first], pt_cnt[p, first], pt_idx[p, first], pt_val[p, first]
    return dict(pt_nsrc=pt_nsrc, pt_src=pt_src, pt_rows=pt_rows, pt_cnt=pt_cnt, pt_idx=pt_idx, pt_val=pt_val,
                fetch=n_src / n, rows_per_patch=n / n_p, groups=[rows for rows, _ in patches])


def _cheb2_csr(rowptr: np.ndarray, colidx: np.ndarray, val: np.ndarray, n: int):
    """CSR (int32 rowptr / colidx, float32 values, columns ascending) of T_2(A) = 2 A^2 - I for a square CSR matrix A: the second Chebyshev
    matrix as the reference's ``cheby_poly`` forms it on the matrix side (STC_GNN.py:24-29), products and sums in float64, rounded once.
    For the small-graph cell kernels at order 3 (N * C < 65 536 rows: A^2 of an 8-neighbour grid has 25 entries per row)."""
    rp = rowptr.astype(np.int64)
    ci = colidx.astype(np.int64)
    v = val.astype(np.float64)
    deg = np.diff(rp)
    rows = np.repeat(np.arange(n, dtype=np.int64), deg)                 # row i of every entry (i, k)
    cnt = deg[ci]                                                       # entries of row k, per entry (i, k)
    total = int(cnt.sum())
    pi = np.repeat(rows, cnt)
    first = np.repeat(rp[ci], cnt)
    within = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    e2 = first + within                                                 # the entry (k, j) each product pairs with
    key = np.concatenate([pi * n + ci[e2], np.arange(n, dtype=np.int64) * (n + 1)])
    w = np.concatenate([2.0 * np.repeat(v, cnt) * v[e2], -np.ones(n)])
    uniq, inv = np.unique(key, return_inverse=True)
    vals = np.bincount(inv, weights=w, minlength=uniq.size)
    out_rows = uniq // n
    rp2 = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rp2, out_rows + 1, 1)
    rp2 = np.cumsum(rp2)
    if uniq.size > INT32_MAX:
        raise ValueError('T_2 of the graph does not fit int32 CSR indices')
    return rp2.astype(np.int32), (uniq - out_rows * n).astype(np.int32), vals.astype(np.float32)


RING2_INTERIOR, RING2_FIRST, RING2_SECOND, RING2_WIDTH = 32, 64, 96, 8       # = STC_RING2_* of include/stc_hip.h


def _ring2_groups(rowptr: np.ndarray, colidx: np.ndarray, n: int, min_rows: float = 16.0):
    """Patches for the two-ring form on a graph that is not a lattice in its numbering: grown like the clusters of ``_patch_plan`` (breadth-first
    over the symmetrised pattern from the first row not yet taken), but bounded by what the two-ring kernel stages -- a row joins while the
    first ring stays within RING2_FIRST rows and the second within RING2_SECOND.  On the bench's grid under a random numbering, renumbered by
    reverse Cuthill-McKee: 29.4 rows per patch, 1.98 first-ring and 3.2 staged rows per own row (4 x 8 tiles of the lattice: 32, 1.88, 3.0).
    None when the patches come out small (no locality).  Deterministic in the arrays alone."""
    rp = rowptr.astype(np.int64)
    if n == 0 or colidx.size == 0 or int(np.diff(rp).max()) > RING2_WIDTH:
        return None
    import scipy.sparse as sp                           # optional, as for the patch plan: the caller leaves the two launches in place without it
    rpl, cil = rp.tolist(), colidx.tolist()
    A = sp.csr_matrix((np.ones(colidx.size, dtype=np.int8), colidx, rp), shape=(n, n))
    S = (A + A.T).tocsr()
    S.sort_indices()
    srp, sci = S.indptr.tolist(), S.indices.tolist()
    taken = np.zeros(n, dtype=bool)
    groups = []
    for seed in range(n):
        if taken[seed]:
            continue
        rows, first, second, queue, queued, qi = [], set(), set(), [seed], {seed}, 0
        while qi < len(queue) and len(rows) < RING2_INTERIOR:
            u = queue[qi]
            qi += 1
            grown = first | set(cil[rpl[u]:rpl[u + 1]]) | {u}
            if len(grown) > RING2_FIRST:
                continue                               # (the row stays available to later patches; a single row always fits: width <= 8)
            staged = set(second)
            for v in grown - first:
                staged.update(cil[rpl[v]:rpl[v + 1]])
            if len(staged) > RING2_SECOND:
                continue
            first, second = grown, staged
            rows.append(u)
            taken[u] = True
            for w in sci[srp[u]:srp[u + 1]]:
                if not taken[w] and w not in queued:
                    queued.add(w)
                    queue.append(w)
        groups.append(rows)
    return groups if n / len(groups) >= min_rows else None


def _ring2_plan(rowptr: np.ndarray, colidx: np.ndarray, val: np.ndarray, groups):
    """Two-ring form of a square CSR matrix S over the patches ``groups`` of its patch plan, for ``stc_ring2_sum_f32`` (include/stc_hip.h): per
    patch its own rows, its first ring (own rows + every column they touch, <= 64), the second ring (every column the first ring's rows
    touch, <= 96) and (LDS offset, value) tables of width 8 for both levels; None when a ring or a row does not fit (the two launches remain).
    On the 8-neighbour grid in 4 x 8 tiles: 32 / 60 / 96 rows."""
    rpl, cil = rowptr.astype(np.int64).tolist(), colidx.tolist()
    n_p = len(groups)
    l2 = np.zeros((n_p, RING2_SECOND), dtype=np.int32)
    l1 = np.full((n_p, RING2_FIRST), -1, dtype=np.int32)
    own = np.full((n_p, RING2_INTERIOR), -1, dtype=np.int32)
    t1 = np.zeros((n_p, RING2_FIRST, RING2_WIDTH, 2), dtype=np.int32)
    t2 = np.zeros((n_p, RING2_INTERIOR, RING2_WIDTH, 2), dtype=np.int32)
    bits = np.asarray(val, dtype=np.float32).view(np.int32)
    chunk_bytes = 32 * 16                              # one row of the kernel's LDS tile: 32 float4

    def fill(table, r, a, b, pos):
        k = b - a
        if k:
            table[r, :k, 0] = [pos[c] * chunk_bytes for c in cil[a:b]]
            table[r, :k, 1] = bits[a:b]
            table[r, k:, 0] = table[r, k - 1, 0]        # the tail: zero-valued repeats of the last entry (bits 0 = 0.0f)

    for p, rows in enumerate(groups):
        if len(rows) > RING2_INTERIOR:
            return None
        first = {u: i for i, u in enumerate(rows)}      # slot of every first-ring row: own rows lead
        for u in rows:
            if rpl[u + 1] - rpl[u] > RING2_WIDTH:
                return None
            for c in cil[rpl[u]:rpl[u + 1]]:
                if c not in first:
                    first[c] = len(first)
        if len(first) > RING2_FIRST:
            return None
        second = {}
        for u in first:
            if rpl[u + 1] - rpl[u] > RING2_WIDTH:
                return None
            for c in cil[rpl[u]:rpl[u + 1]]:
                if c not in second:
                    second[c] = len(second)
        if len(second) > RING2_SECOND:
            return None
        l2[p] = next(iter(second), 0)
        l2[p, :len(second)] = list(second)
        for u, slot in first.items():
            l1[p, slot] = u | (1 << 30) if slot < len(rows) else u
            fill(t1[p], slot, rpl[u], rpl[u + 1], second)
        for r, u in enumerate(rows):
            own[p, r] = u
            fill(t2[p], r, rpl[u], rpl[u + 1], first)
    return dict(r2_l2=l2, r2_l1=l1, r2_own=own, r2_t1=t1, r2_t2=t2)


class CsrGraph:
    """A fixed N x N spatial graph ``Gs`` resident in HBM as CSR(Gs^T) + CSR(Gs).

    ``fwd_*`` describe Gs^T (row m lists the nodes n with Gs[n, m] != 0): the operand of the
    forward aggregation.  ``bwd_*`` describe Gs itself, used for dX.  ``bwd_perm`` maps the
    backward value order onto the forward one (``bwd_val = fwd_val[bwd_perm]``).
    """

    def __init__(self, n: int, rows, cols, vals, device=None):
        rows = np.asarray(rows, dtype=np.int64).ravel()
        cols = np.asarray(cols, dtype=np.int64).ravel()
        vals = np.asarray(vals, dtype=np.float32).ravel()
        if not (rows.shape == cols.shape == vals.shape):
            raise ValueError('rows / cols / vals must have the same length')
        if n < 0 or n > INT32_MAX or rows.size > INT32_MAX:
            raise ValueError('graph too large for int32 CSR indices')
        if rows.size and (rows.min() < 0 or rows.max() >= n or cols.min() < 0 or cols.max() >= n):
            raise ValueError(f'edge index outside [0, {n})')
        if rows.size:
            key = rows * n + cols
            if np.unique(key).size != key.size:
                raise ValueError('duplicate (row, col) entries: coalesce the graph first')
        self.n = int(n)
        self.nnz = int(rows.size)
        # forward operand Gs^T: entry (n_, m) of Gs is stored in row m, column n_
        f_rp, f_ci, f_v, f_order = _csr_from_coo(cols, rows, vals, n)
        b_rp, b_ci, b_v, b_order = _csr_from_coo(rows, cols, vals, n)
        inv_f = np.empty_like(f_order)
        inv_f[f_order] = np.arange(f_order.size)
        perm = inv_f[b_order]                      # position in fwd order of each bwd entry
        self._host = dict(fwd_rowptr=f_rp, fwd_colidx=f_ci, fwd_val=f_v,
                          bwd_rowptr=b_rp, bwd_colidx=b_ci, bwd_val=b_v, bwd_perm=perm.astype(np.int64))
        #: max over both orientations of the largest absolute row sum: what ONE aggregation can amplify a state (and its rounding noise) by.
        #: Row-stochastic graphs: 1; the reference's raw 0/1 adjacency: 8.  ``ops`` routes graphs beyond ``HEAVY_ROW_SUM`` to the 24-bit
        #: operand format (see there).
        self.row_sum_bound = float(max((np.add.reduceat(np.abs(v), rp[:-1][np.diff(rp) > 0]).max() if v.size else 0.0)
                                       for rp, v in ((f_rp.astype(np.int64), f_v), (b_rp.astype(np.int64), b_v))))
        distinct = {}
        #: side -> (source rows per output row, rows per patch) of the patch form, where the graph has one
        self.patch_stats: Dict[str, Tuple[float, float]] = {}
        #: the two-ring plans are over ring-bounded clusters (a graph that is not a lattice in its numbering), not over the patch form's patches
        self.ring2_clusters = False
        for side, (rp, ci, v) in (('fwd', (f_rp, f_ci, f_v)), ('bwd', (b_rp, b_ci, b_v))):
            plan = _row_block_plan(rp, ci, v, n)
            distinct[side] = plan.pop('distinct')
            self._host.update({f'{side}_{k}': a for k, a in plan.items()})
            try:
                patch = _patch_plan(rp, ci, v, n)
            except ImportError:                         # scipy is optional (as for the locality order): no patch form without it
                patch = None
            if patch is not None:
                self.patch_stats[side] = (patch.pop('fetch'), patch.pop('rows_per_patch'))
                groups = patch.pop('groups')
                self._host.update({f'{side}_{k}': a for k, a in patch.items()})
                # two-ring plans: the forward's blend + aggregation of the new state (Gs^T, stc_ring2_blend_f32) and the state-gradient path (Gs,
                # stc_ring2_sum_f32)
                ring2 = _ring2_plan(rp, ci, v, groups)
                if ring2 is None:                       # clusters of the patch form whose rings do not fit: patches grown for the rings instead
                    try:
                        own = _ring2_groups(rp, ci, n)
                    except ImportError:                 # scipy is optional here too: no ring-bounded clusters, the two launches remain
                        own = None
                    ring2 = _ring2_plan(rp, ci, v, own) if own is not None else None
                    self.ring2_clusters = self.ring2_clusters or ring2 is not None
                if ring2 is not None:
                    self._host.update({f'{side}_{k}': a for k, a in ring2.items()})
        #: distinct neighbour rows fetched per output row by the row-blocked kernel (CSR: nnz / n)
        self.fetches_per_row = tuple(distinct[s_] / max(n, 1) for s_ in ('fwd', 'bwd'))
        self._dev: Dict[torch.device, Dict[str, torch.Tensor]] = {}
        if device is not None:
            self.on(torch.device(device))

    # -- constructors -------------------------------------------------------------------
    @classmethod
    def from_dense(cls, G, device=None) -> 'CsrGraph':
        G = G.detach().cpu() if isinstance(G, torch.Tensor) else torch.as_tensor(np.asarray(G))
        if G.dim() != 2 or G.shape[0] != G.shape[1]:
            raise ValueError('from_dense wants a square matrix')
        idx = G.nonzero(as_tuple=False)
        return cls(G.shape[0], idx[:, 0].numpy(), idx[:, 1].numpy(), G[idx[:, 0], idx[:, 1]].numpy(), device)

    @classmethod
    def from_torch_sparse(cls, S: torch.Tensor, device=None) -> 'CsrGraph':
        if S.dim() != 2 or S.shape[0] != S.shape[1]:
            raise ValueError('sparse spatial graph must be square')
        coo = S.detach().cpu().to_sparse_coo().coalesce()
        i = coo.indices().numpy()
        return cls(S.shape[0], i[0], i[1], coo.values().numpy(), device if device is not None else S.device)

    @classmethod
    def queen_grid(cls, H: int, W: int, normalize: bool = True, permute_seed: Optional[int] = None,
                   device=None) -> 'CsrGraph':
        """H x W 8-neighbour grid, node id h*W + w (the SF ``s_adj`` for 10 x 10); optionally
        row-stochastic (A / rowsum) and under a seeded random node permutation."""
        hh, ww = np.divmod(np.arange(H * W, dtype=np.int64), W)
        rows, cols = [], []
        for dh in (-1, 0, 1):
            for dw in (-1, 0, 1):
                if dh == 0 and dw == 0:
                    continue
                nh, nw = hh + dh, ww + dw
                ok = (nh >= 0) & (nh < H) & (nw >= 0) & (nw < W)
                rows.append((hh * W + ww)[ok])
                cols.append((nh * W + nw)[ok])
        r, c = np.concatenate(rows), np.concatenate(cols)
        v = np.ones(r.size, dtype=np.float32)
        if normalize:
            deg = np.bincount(r, minlength=H * W).astype(np.float32)
            v = v / deg[r]
        if permute_seed is not None:
            g = torch.Generator().manual_seed(permute_seed)
            p = torch.randperm(H * W, generator=g).numpy()      # new matrix A[p][:, p]: new id i is old node p[i]
            inv = np.empty_like(p)
            inv[p] = np.arange(p.size)
            r, c = inv[r], inv[c]
        return cls(H * W, r, c, v, device)

    # -- locality ------------------------------------------------------------------------------
    def locality_order(self) -> np.ndarray:
        """Reverse Cuthill-McKee order of the symmetrised pattern: new node i is old node order[i].

        The SpMM gathers whole feature rows; what it costs is set by how many DISTINCT neighbour rows a block of
        consecutive output rows touches and whether they are still in the XCD's L2.  A graph handed over in an
        arbitrary node order (e.g. the queen grid under a random permutation: 7.9 fetches per row, SpMM at 23 % of
        HBM peak) is brought back to a banded form (4.5 fetches per row, 50 %) by renumbering the nodes once.
        """
        import scipy.sparse as sp
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        h = self._host
        A = sp.csr_matrix((np.ones(self.nnz, dtype=np.int8), h['bwd_colidx'], h['bwd_rowptr']), shape=(self.n, self.n))
        return np.asarray(reverse_cuthill_mckee((A + A.T).tocsr(), symmetric_mode=True), dtype=np.int64)

    def permuted(self, order: np.ndarray) -> 'CsrGraph':
        """The same graph with nodes renumbered: G'[i, j] = G[order[i], order[j]]."""
        order = np.asarray(order, dtype=np.int64)
        inv = np.empty_like(order)
        inv[order] = np.arange(order.size)
        h = self._host
        rows = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(h['bwd_rowptr'].astype(np.int64)))
        return CsrGraph(self.n, inv[rows], inv[h['bwd_colidx'].astype(np.int64)], h['bwd_val'])

    def _locality_order_once_per_job(self):
        """``locality_order()`` computed by rank 0 and broadcast when a process group is up (one process per GPU: eight ranks would otherwise
        run the same host-side analysis side by side and -- worse -- could disagree if their scipy builds differed: every rank must renumber
        the nodes the same way, or the replicated graph is no longer the same graph).  Collective on the default group: every rank calls it, at
        the same point (the first forward pass of a graph) -- a graph only SOME ranks use must be renumbered beforehand, or not at all:
        ``CsrGraph.with_locality(collective=False)`` computes the order locally.  None: scipy is missing on rank 0 (the given order is kept
        everywhere); any other failure on rank 0 is raised on every rank."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            try:
                return self.locality_order()
            except ImportError:                     # scipy (reverse Cuthill-McKee) is optional: without it the given order is kept
                return None
        on_gpu = dist.get_backend() == 'nccl'
        buf = torch.zeros(self.n + 1, dtype=torch.int64)
        failure = None
        if dist.get_rank() == 0:
            try:
                buf[1:] = torch.from_numpy(self.locality_order())
                buf[0] = 1
            except ImportError:
                pass
            except Exception as e:                  # whatever goes wrong here must not leave the other ranks waiting in the broadcast
                failure = e
                buf[0] = 2
        if on_gpu:
            buf = buf.cuda()
        dist.broadcast(buf, src=0)
        buf = buf.cpu()
        status = int(buf[0])
        if status == 2:                             # every rank raises: rank 0 its own exception, the others what they were told
            raise failure if failure is not None else RuntimeError('CsrGraph: rank 0 failed to compute the node renumbering (see its log)')
        return buf[1:].numpy().copy() if status == 1 else None

    def with_locality(self, min_gain: float = 1.25, collective: bool = True):
        """(graph, order): a renumbered copy when that cuts the row fetches of the row-blocked SpMM by at least
        ``min_gain``, else (self, None).  Computed once and cached.  Under a process group the order comes from rank 0 (a collective on the
        default group, see above); ``collective=False`` computes it in this process alone -- for a graph that not every rank uses (rank-0-only
        evaluation on a new graph): call it once before the first forward pass."""
        cached = getattr(self, '_locality', None)
        if cached is None:
            cached = (self, None)
            if self.n > 64 and self.nnz > 0:
                if collective:
                    order = self._locality_order_once_per_job()
                else:
                    try:
                        order = self.locality_order()
                    except ImportError:
                        order = None
                if order is None:
                    import warnings
                    warnings.warn('scipy is not installed: the spatial graph keeps its node order (no locality renumbering)')
                if order is not None:
                    cand = self.permuted(order)
                    if self.fetches_per_row[0] >= min_gain * cand.fetches_per_row[0]:
                        cached = (cand, order)
            self._locality = cached
        return cached

    # -- device residency -----------------------------------------------------------------
    def on(self, device: torch.device) -> Dict[str, torch.Tensor]:
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        d = self._dev.get(device)
        if d is None:
            d = {k: torch.from_numpy(v).to(device) for k, v in self._host.items()}
            self._dev[device] = d
        return d

    def second_order(self, device: torch.device) -> Dict[str, torch.Tensor]:
        """CSR of the second Chebyshev matrix T_2 = 2 S^2 - I in both orientations, on ``device`` (cached): ``fwd2_*`` = 2 (Gs^T)^2 - I for the
        forward's aggregation, ``bwd2_*`` = 2 Gs^2 - I for its transpose -- what the small-graph cell kernels take as their second graph at
        Chebyshev order 3 (the reference forms T_2 on the matrix side as well: STC_GNN.py:24-29)."""
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        cache = self.__dict__.setdefault('_dev2', {})
        d = cache.get(device)
        if d is None:
            h, d = self._host, {}
            for side in ('fwd', 'bwd'):
                rp, ci, v = _cheb2_csr(h[f'{side}_rowptr'], h[f'{side}_colidx'], h[f'{side}_val'], self.n)
                d.update({f'{side}2_rowptr': torch.from_numpy(rp).to(device), f'{side}2_colidx': torch.from_numpy(ci).to(device),
                          f'{side}2_val': torch.from_numpy(v).to(device)})
            cache[device] = d
        return d

    def to_dense(self) -> torch.Tensor:
        h = self._host
        G = torch.zeros(self.n, self.n)
        rows = np.repeat(np.arange(self.n), np.diff(h['bwd_rowptr']))
        G[torch.from_numpy(rows), torch.from_numpy(h['bwd_colidx'].astype(np.int64))] = torch.from_numpy(h['bwd_val'])
        return G

    def __repr__(self):
        return f'CsrGraph(n={self.n}, nnz={self.nnz})'


@dataclass
class SpatialOperand:
    """What one BDG_Dif call needs of the spatial graph (either mode)."""
    n: int
    fwd_rowptr: torch.Tensor
    fwd_colidx: torch.Tensor
    fwd_val: torch.Tensor            # may carry autograd history (learned dense graph)
    bwd_rowptr: torch.Tensor
    bwd_colidx: torch.Tensor
    bwd_val: torch.Tensor            # never differentiable
    nnz: int
    # (blk_ptr, blk_cols, blk_vals) of a fixed graph, plus -- where its rows cluster -- a fourth item: the patch form
    # (pt_src, pt_rows, pt_cnt, pt_idx, pt_val) of stc_patch_spmm_f32
    fwd_plan: Optional[tuple] = None
    bwd_plan: Optional[tuple] = None
    row_sum_bound: float = 1.0       # fixed graphs: max absolute row sum over both orientations (CsrGraph.row_sum_bound)
    bwd_ring2: Optional[tuple] = None    # (l2_rows, l1_rows, int_rows, t1, t2) of stc_ring2_sum_f32 for Gs, where its patches' rings fit
    fwd_ring2: Optional[tuple] = None    # ... of stc_ring2_blend_f32 for Gs^T
    ring2_clusters: bool = False         # those plans are over ring-bounded clusters (1 708 ragged patches for the bench's grid under a random order
                                         # against 1 568 tiles: every launch +17 %), where only the forms that replace two gathering launches pay
    source: Optional['CsrGraph'] = None  # the fixed graph this operand came from (its second Chebyshev matrix: CsrGraph.second_order)


_PATTERN_CACHE: Dict[Tuple[int, torch.device], Tuple[torch.Tensor, torch.Tensor]] = {}
_PATTERN_PTRS: Dict[int, int] = {}          # address of a cached colidx -> its n


def full_pattern(n: int, device: torch.device):
    """rowptr / colidx of the dense n x n pattern (cached per size and device)."""
    device = torch.device(device)
    key = (n, device)
    got = _PATTERN_CACHE.get(key)
    if got is None:
        if n * n > INT32_MAX:
            raise ValueError(f'dense spatial graph of {n} nodes does not fit int32 CSR; use a CsrGraph')
        rowptr = (torch.arange(n + 1, dtype=torch.int64) * n).to(torch.int32).to(device)
        colidx = torch.arange(n, dtype=torch.int32).repeat(n).to(device)
        got = (rowptr, colidx)
        _PATTERN_CACHE[key] = got
        _PATTERN_PTRS[colidx.data_ptr()] = n
    return got


def is_full_pattern(colidx: torch.Tensor, n_rows: int, n_cols: int) -> bool:
    """Whether ``colidx`` IS the cached column array of the dense pattern (``full_pattern``): the values that go with it are then a dense
    row-major (n_rows, n_cols) matrix, and the aggregation is a dense product (``stc_dense_agg_f32``)."""
    if n_rows != n_cols or colidx.numel() != n_rows * n_cols:
        return False
    return _PATTERN_PTRS.get(colidx.data_ptr()) == n_cols          # (the cache keeps the array alive, so the address cannot be reused)


def dense_operand(Gs: torch.Tensor) -> SpatialOperand:
    """The reference's learned dense ``Gs`` (N,N) as a full-pattern CSR operand.

    The forward values are ``Gs^T`` flattened (differentiable: autograd carries the transpose
    back to ``Gs``), the backward values ``Gs`` itself.
    """
    if Gs.dim() != 2 or Gs.shape[0] != Gs.shape[1]:
        raise ValueError(f'Gs must be (N, N), got {tuple(Gs.shape)}')
    n = Gs.shape[0]
    rowptr, colidx = full_pattern(n, Gs.device)
    fwd_val = Gs.t().contiguous().reshape(-1)
    bwd_val = Gs.detach().contiguous().reshape(-1)
    return SpatialOperand(n, rowptr, colidx, fwd_val, rowptr, colidx, bwd_val, n * n)


def csr_operand(graph: CsrGraph, device: torch.device) -> SpatialOperand:
    d = graph.on(device)
    def plan(side):
        blocks = (d[f'{side}_blk_ptr'], d[f'{side}_blk_cols'], d[f'{side}_blk_vals'])
        if f'{side}_pt_src' not in d:
            return blocks
        return blocks + (tuple(d[f'{side}_pt_{k}'] for k in ('src', 'rows', 'cnt', 'idx', 'val')),)
    return SpatialOperand(graph.n, d['fwd_rowptr'], d['fwd_colidx'], d['fwd_val'],
                          d['bwd_rowptr'], d['bwd_colidx'], d['bwd_val'], graph.nnz, plan('fwd'), plan('bwd'), graph.row_sum_bound,
                          *[tuple(d[f'{sd}_r2_{k}'] for k in ('l2', 'l1', 'own', 't1', 't2')) if f'{sd}_r2_l2' in d else None for sd in ('bwd', 'fwd')],
                          graph.ring2_clusters, graph)
