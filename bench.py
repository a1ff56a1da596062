#!/usr/bin/env python3
"""Benchmark of the STC-GNN hot path on MI355X: BASELINE.json's metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One "step" = one train step of the full encoder-decoder STC-GNN in ``csr-fixed`` mode on synthetic
tensors already resident in HBM: forward, ComboLoss, backward, the RCCL all-reduce of the gradient
bucket (N > 1) and the Adam update.  Workload: 224x224 row-stochastic queen grid
(N = 50 176 nodes, nnz = 398 724), C = 32 categories, hidden 16, 2 layers, Chebyshev order K,
T = 18 observed + 6 predicted steps, Bernoulli(0.1635) inputs, batch per GPU fixed (weak scaling:
the batch x time-window dimension is sharded, the graph replicated, no data-path collective).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the CSR-SpMM aggregation (the kernel the metric names): algorithmic bytes per launch
                (nnz*(4+4) + 4*(N+1) + 2*B*N*F*4; + B*N*F*4 for the launches whose epilogue
                also reads Y0, i.e. Y = Y0 + S.X of the backward) / mean launch duration, measured with
                HIP events on the launching stream around every SpMM launch of the timed steps
  kernels       time share of every C-ABI entry point over the same steps (same events)
  cpu_baseline  the CPU oracle (oracle/stc_oracle.py, sparse feature-side variant because the
                reference's dense N x N formulation needs 20 GB at this N) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, 'stc-gnn_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6300
# every C-ABI entry point that is an aggregation Y = S.X (+ epilogue): the kernels the metric's roofline is about
SPMM_ENTRY_POINTS = ('stc_bcsr_spmm_f32', 'stc_csr_spmm_f32', 'stc_spmm_sum_f32', 'stc_spmm_blend_fwd_f32',
                     'stc_bcsr_spmm_bf16', 'stc_csr_spmm_bf16', 'stc_spmm_sum_bf16', 'stc_spmm_blend_fwd_bf16')
METRIC = 'STC-GNN fwd+bwd samples/sec at N=50k,C=32; SpMM HBM GB/s vs peak, 1/8 GPU'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--grid', type=int, default=224, help='H = W of the queen grid (N = grid^2)')
    ap.add_argument('--categories', type=int, default=32)
    ap.add_argument('--hidden', type=int, default=16)
    ap.add_argument('--order', type=int, default=2, help='Chebyshev order Ks = Kc')
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--obs', type=int, default=18)
    ap.add_argument('--pred', type=int, default=6)
    ap.add_argument('--batch-per-gpu', type=int, default=5, help='samples per GPU (weak scaling); 38 GB of saved activations per sample: 5 = 189 GB of the 288 GB (6 fits too: 227 GB)')
    ap.add_argument('--permute', action='store_true', help='random node order (seed 1234) instead of row-major')
    ap.add_argument('--no-reorder', action='store_true', help='keep the given node order (skip the internal RCM renumbering)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--storage', choices=('f32', 'bf16'), default='f32',
                    help='storage type of states / gates / their gradients: f32 = the metric (reference arithmetic); bf16 = BASELINE configuration 5 '
                         '(not the metric: bf16 has no reference behaviour), fp32 parameters and fp32 sums inside every kernel')
    return ap.parse_args()


def cpu_baseline(a, graph_dense_sparseT, Gc, sd_cpu):
    """Oracle fwd+bwd on the host cores for a bounded sample of the same workload.

    Sample: the same graph/width, batch 1, but 1 observed + 1 predicted step (4 of the 2*(obs+pred)
    cell evaluations of a full sample); samples/s is scaled by the cell count.  kind = "port": the
    oracle's sparse restatement (the reference itself cannot run N = 50 176: dense Gs + eye = 20 GB
    and an N^3 cheby_poly), validated against the dense reference at N = 1 024 / 10 000 in tests/.
    """
    from oracle import stc_oracle as O
    threads = min(os.cpu_count() or 1, 32)      # more threads than this only slow the torch CPU ops down on a 2-socket host
    torch.set_num_threads(threads)
    N, C = a.grid * a.grid, a.categories
    g = torch.Generator().manual_seed(0)
    X = (torch.rand(1, 1, N, C, generator=g) < 0.1635).float()
    Y = (torch.rand(1, 1, N, C, generator=g) < 0.1635).float()
    sd = {k: v.clone().requires_grad_() for k, v in sd_cpu.items()}
    t0 = time.perf_counter()
    yhat = O.encdec_forward(X, graph_dense_sparseT, Gc, sd, a.order, a.order, a.hidden, a.layers, 1,
                            conv=O.bdg_dif_sparse)
    O.combo_loss(yhat, Y).backward()
    dt = time.perf_counter() - t0
    cells_sample = a.layers * 2
    cells_full = a.layers * (a.obs + a.pred)
    return dict(value=1.0 / (dt * cells_full / cells_sample), unit='samples/s', cores=threads, kind='port',
                sample=f'oracle (torch CPU, sparse feature-side variant) fwd+bwd, batch 1, same graph/width, '
                       f'1 obs + 1 pred step = {cells_sample} of {cells_full} cell evaluations, {dt:.1f} s; scaled by cell count')


def main():
    a = parse()
    from stc_hip import CsrGraph, ops
    from stc_hip import dist as sdist
    from stc_hip._lib import KernelTimer
    from stc_hip.loss import ComboLoss
    import STC_GNN as M

    rank, world, local = sdist.init_from_env()
    if world != a.gpus:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}')
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    N, C, B = a.grid * a.grid, a.categories, a.batch_per_gpu
    graph = CsrGraph.queen_grid(a.grid, a.grid, normalize=True, permute_seed=1234 if a.permute else None, device=dev)
    Gc_cpu = torch.softmax(torch.randn(C, C, generator=torch.Generator().manual_seed(7)), -1)
    torch.manual_seed(42)                                                       # same parameters on every rank
    model = M.STCGNN(N, C, a.order, a.order, 1, a.hidden, a.layers, a.pred, graph_mode='csr-fixed', reorder_nodes=not a.no_reorder,
                     storage_dtype=torch.bfloat16 if a.storage == 'bf16' else torch.float32)
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    Gc = Gc_cpu.to(dev)
    g = torch.Generator().manual_seed(1000 + rank)                              # a different shard of samples per rank
    X = (torch.rand(B, a.obs, N, C, generator=g) < 0.1635).float().to(dev)
    Y = (torch.rand(B, a.pred, N, C, generator=g) < 0.1635).float().to(dev)
    crit = ComboLoss()
    bucket = sdist.GradBucket(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=2e-3, weight_decay=1e-4)

    def step():
        bucket.zero()
        loss = crit(model(X_seq=X, As=graph, Ac=Gc), Y)
        loss.backward()
        bucket.allreduce_mean()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    hip = ops.kernels()
    fence()
    hip.timer = KernelTimer()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    per_kernel = hip.timer.summary()
    hip.timer = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    if rank == 0:
        total_ms = sum(d['ms'] for d in per_kernel.values()) or 1.0
        spmm = dict(launches=0, ms=0.0, bytes=0)           # both forms of the aggregation: CSR and row-blocked CSR
        for name in SPMM_ENTRY_POINTS:
            for key, v in per_kernel.get(name, {}).items():
                spmm[key] += v
        achieved = (spmm['bytes'] / 1e9) / (spmm['ms'] / 1e3) if spmm['ms'] > 0 else 0.0
        # HBM bytes per SpMM launch from the committed PMC passes over this same command (FETCH_SIZE x 2 + WRITE_SIZE in
        # separate rocprofv3 --pmc runs, as MI355X_MICROARCH.md prescribes; tools/gpu_pmc_bench.sh): averaged over all SpMM
        # launches of a step exactly like roofline.achieved.  Only valid for the configuration it was collected on.
        traffic, traffic_note = None, None
        tpath = os.path.join(REPO, 'profiles', 'r01', 'j_hbm_traffic_bench_b5.json')
        if a.storage == 'bf16' and (a.grid, C, a.hidden, B, a.order, a.layers, a.obs, a.pred, a.permute) == (224, 64, 16, 5, 2, 2, 18, 6, False):
            tb = os.path.join(REPO, 'profiles', 'r01', 'k_hbm_traffic_bench_bf16_c64.json')       # the same passes over --storage bf16 --categories 64
            if os.path.exists(tb):
                with open(tb) as fh:
                    ks = [v for name, v in json.load(fh)['kernels'].items() if name.startswith('spmm_')]
                if ks:
                    traffic = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in ks) / sum(v['launches'] for v in ks)
                    traffic_note = 'PMC (2 x FETCH_SIZE + WRITE_SIZE), mean over the SpMM launches of one step of this command: profiles/r01/k_hbm_traffic_bench_bf16_c64.json'
        if a.storage == 'f32' and (a.grid, C, a.hidden, B, a.order, a.layers, a.obs, a.pred, a.permute) == (224, 32, 16, 5, 2, 2, 18, 6, False) and os.path.exists(tpath):
            with open(tpath) as fh:
                doc = json.load(fh)
            ks = [v for name, v in doc['kernels'].items() if name.startswith('spmm_')]
            if ks:
                traffic = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in ks) / sum(v['launches'] for v in ks)
                traffic_note = 'PMC (2 x FETCH_SIZE + WRITE_SIZE), mean over the SpMM launches of one step of this command: profiles/r01/j_hbm_traffic_bench_b5.json'
        out = {
            'metric': METRIC, 'value': world * B * a.steps / elapsed, 'unit': 'samples/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * elapsed / a.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.storage, 'data': 'synthetic',
            'config': {'workload': f'full STC-GNN train step (fwd + ComboLoss + bwd + grad all-reduce + Adam), csr-fixed, '
                                   f'{a.grid}x{a.grid} queen grid N={N} nnz={graph.nnz}{" permuted" if a.permute else ""}, C={C}, '
                                   f'hidden={a.hidden}, K={a.order}, layers={a.layers}, T={a.obs}+{a.pred}' + (', bf16 state storage' if a.storage == 'bf16' else ''),
                       'global_batch': world * B, 'batch_per_gpu': B, 'parallelism': f'batch-shard x{world}',
                       'grad_bucket_bytes': bucket.nbytes},
            'roofline': {'bound': 'hbm', 'kernel': ' + '.join(SPMM_ENTRY_POINTS) + ': every aggregation (SpMM) launch of the timed steps -- plain, with the GRU blend in its epilogue, and the state-gradient sum form', 'achieved': achieved, 'peak': HBM_PEAK_GBPS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic, 'traffic_note': traffic_note,
                         'launches': spmm['launches'],
                         'avg_launch_us': 1e3 * spmm['ms'] / max(1, spmm['launches']),
                         'algorithmic_bytes_per_launch': spmm['bytes'] / max(1, spmm['launches'])},
            'kernels': {k: {'launches': d['launches'], 'ms_per_step': d['ms'] / a.steps, 'share': d['ms'] / total_ms,
                            **({'GBps': d['bytes'] / 1e9 / (d['ms'] / 1e3)} if d['bytes'] and d['ms'] > 0 else {})}
                        for k, d in sorted(per_kernel.items(), key=lambda kv: -kv[1]['ms'])},
            'loss': float(loss.detach()),
            'hbm_peak_allocated_gb': torch.cuda.max_memory_allocated(dev) / 1e9,
            'hbm_peak_reserved_gb': torch.cuda.max_memory_reserved(dev) / 1e9,
        }
        if world == 1 and not a.no_cpu_baseline:
            GsT = graph.to_dense().t().contiguous().to_sparse_csr() if N <= 4096 else _sparse_T(graph)
            out['cpu_baseline'] = cpu_baseline(a, GsT, Gc_cpu, sd_cpu)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _sparse_T(graph):
    """torch sparse CSR of Gs^T straight from the graph's forward operand (no dense N x N detour)."""
    h = graph._host
    return torch.sparse_csr_tensor(torch.from_numpy(h['fwd_rowptr']).long(), torch.from_numpy(h['fwd_colidx']).long(),
                                   torch.from_numpy(h['fwd_val']), size=(graph.n, graph.n))


if __name__ == '__main__':
    main()
