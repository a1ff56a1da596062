"""The plain aggregation unit of the bench (224 x 224 queen grid, B = 5, F = 512 floats; KNN=1: a k-nearest-neighbour mesh of as many random
points) through the row-blocked and the patch kernel,
alone on the GPU (alternating buffer pairs, HIP events around 30 launches):  python tools/probes/patch_spmm_unit.py [grid] [B] [F] [permute_seed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'stc-gnn_amd'))
from stc_hip import CsrGraph          # noqa: E402
from stc_hip._lib import HipKernels   # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 224
B = int(sys.argv[2]) if len(sys.argv) > 2 else 5
F = int(sys.argv[3]) if len(sys.argv) > 3 else 512
seed = int(sys.argv[4]) if len(sys.argv) > 4 else None
hip = HipKernels()
if os.environ.get('PATCH_MIN') is not None:
    hip.patch_min_items = int(os.environ['PATCH_MIN'])      # the launch-size rule of _lib.py (default 3 072 / 6 144 items)
if os.environ.get('KNN'):                         # an irregular mesh instead of the grid: G*G random points in the plane, 8 nearest neighbours each
    import numpy as np
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(0)
    pts = rng.random((G * G, 2))
    idx = cKDTree(pts).query(pts, k=9)[1]
    key = np.unique(np.repeat(np.arange(G * G, dtype=np.int64), 8) * (G * G) + idx[:, 1:].ravel())
    if os.environ.get('KNN') == 'sym':             # ... made symmetric (an undirected mesh)
        key = np.unique(np.concatenate([key, (key % (G * G)) * (G * G) + key // (G * G)]))
    graph = CsrGraph(G * G, key // (G * G), key % (G * G), np.full(key.size, 0.125, dtype=np.float32))
else:
    graph = CsrGraph.queen_grid(G, G, permute_seed=seed)
if os.environ.get('LOCALITY'):
    graph = graph.with_locality()[0]                  # renumbered (reverse Cuthill-McKee), as the model does with a graph in arbitrary order
print('patch plan:', graph.patch_stats, ' row-blocked fetches per row:', graph.fetches_per_row)
if os.environ.get('TILES'):                       # the patches as TY x TX tiles of the grid instead of the clusters (natural node order only)
    import numpy as np
    from stc_hip.graph import _patch_tables
    TY, TX = (int(v) for v in os.environ['TILES'].split('x'))
    h = graph._host
    for side in ('fwd', 'bwd'):
        rpl, cil = h[f'{side}_rowptr'].tolist(), h[f'{side}_colidx'].tolist()
        patches = []
        for ty in range(0, G, TY):
            for tx in range(0, G, TX):
                rows = [y * G + x for y in range(ty, min(ty + TY, G)) for x in range(tx, min(tx + TX, G))]
                src = {}
                for c in sorted({c for u in rows for c in cil[rpl[u]:rpl[u + 1]]}):
                    src[c] = len(src)
                patches.append((rows, src))
        t = _patch_tables(rpl, cil, h[f'{side}_val'], graph.n, patches, 8)
        t.pop('fetch'), t.pop('rows_per_patch')
        h.update({f'{side}_{k}': a for k, a in t.items()})
    print('tiles', TY, 'x', TX, ':', len(patches), 'patches')
d = graph.on(torch.device('cuda'))
n = graph.n
Xs = [torch.zeros(B, n, F, device='cuda') if os.environ.get('ZEROS') else torch.randn(B, n, F, device='cuda') for _ in range(2)]      # ZEROS=1: all-zero operands
Ys = [torch.empty(B, n, F, device='cuda') for _ in range(2)]
def copy_rate():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(3):
        Ys[i & 1].copy_(Xs[i & 1])
    e0.record()
    for i in range(30):
        Ys[i & 1].copy_(Xs[i & 1])
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 30
    print(f'torch copy of the same two planes       {us:7.1f} us  {2 * 4 * B * n * F / us / 1e6:5.2f} TB/s', flush=True)


copy_rate()
for side in ('fwd', 'bwd'):
    if f'{side}_pt_src' not in d:
        print(f'{side}: no patch plan for this orientation')
    blocks = (d[f'{side}_blk_ptr'], d[f'{side}_blk_cols'], d[f'{side}_blk_vals'])
    forms = {'row-blocked': blocks}
    if f'{side}_pt_src' in d:
        forms['patch'] = blocks + (tuple(d[f'{side}_pt_{k}'] for k in ('src', 'rows', 'cnt', 'idx', 'val')),)
    for name, plan in forms.items():
        for y0 in (False, True):
            def go(i):
                hip.csr_spmm(d[f'{side}_rowptr'], d[f'{side}_colidx'], d[f'{side}_val'], n, n, Xs[i & 1], Ys[i & 1] if y0 else None, Ys[i & 1], 1.0, 1.0 if y0 else 0.0, plan=plan)
            for i in range(4):
                go(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(30):
                go(i)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 30
            planes = 3 if y0 else 2
            print(f'{side} {name:12s} {"Y += S.X" if y0 else "Y = S.X ":9s} {us:7.1f} us  {planes * 4 * B * n * F / us / 1e6:5.2f} TB/s  {planes * 4 * B * n * F / us / 8e6:.3f} of peak', flush=True)
