#!/usr/bin/env python3
"""Benchmark of the STC-GNN hot path on MI355X: BASELINE.json's metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W

N > 1 either comes in already launched by ``torch.distributed.run`` (one rank per GPU, RANK / WORLD_SIZE in
the environment) or -- plain ``python bench.py --gpus N`` -- launches itself that way: before anything
touches the GPU this process starts ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a
CHILD, relays its output (rank 0's single JSON line) and exits with its code.  It never exec()s.

One "step" = one train step of the full encoder-decoder STC-GNN in ``csr-fixed`` mode on synthetic
tensors already resident in HBM: forward, ComboLoss, backward, the RCCL all-reduce of the gradient
bucket (N > 1) and the Adam update.  Workload: 224x224 row-stochastic queen grid
(N = 50 176 nodes, nnz = 398 724), C = 32 categories, hidden 16, 2 layers, Chebyshev order K,
T = 18 observed + 6 predicted steps, Bernoulli(0.1635) inputs, batch per GPU fixed (weak scaling:
the batch x time-window dimension is sharded, the graph replicated, no data-path collective).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the PLAIN aggregation launch Y = S.X of the train step (entry point stc_patch_spmm_f32, device kernel
                spmm_patch_kernel; stc_bcsr_spmm_* / spmm_bcsr_kernel for bf16 storage and graphs without a patch plan):
                SURVEY 8(d3)'s algorithmic bytes nnz*(4+4) + 4*(N+1) + 2*B*N*F*4 per launch / its mean
                duration from HIP events recorded on the launching stream around every such launch of the TIMED steps.
                ``aggregate`` beside it = every aggregation launch of the step (plain, GRU blend in the epilogue,
                state-gradient sums) with every operand counted once; ``unit_d3`` = the exact 8(d3) unit (B = 1,
                F = C*L = 1024) timed cold, >= 50 launches, in this same process after the timed steps
  step_breakdown  forward+backward / all-reduce / Adam milliseconds per step (HIP events), and what the per-launch
                event records cost (the same step re-timed with the timer off)
                ``dominant`` = the entry point with the largest share of the step (the one-launch cell backward), its algorithmic
                plane bytes / its mean launch duration; ``mfma`` = matrix-pipe / vector-pipe busy fractions of the projection
                kernels from the committed PMC pass (profiles/rNN/mfma_util.json, newest round), quoted only for the same kernel sources
  kernels       time share of every C-ABI entry point: the priced ones (plain aggregation, dominant cell kernels) over the timed steps, the
                rest from two further steps with every launch timed, scaled to the step count
  cpu_baseline  the CPU oracle (oracle/stc_oracle.py) on a bounded sample: 2 warm-up + 7 timed shots, median

``--preset cfg4 | cfg5 | sf`` selects BASELINE.json's other configurations (same schema, not the metric); ``--preset cfg2`` times the single
BDG_Dif layer of configuration 2.  ``--global-batch G`` fixes the total batch (strong scaling: G / N samples per GPU).
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~5.4-6.3 TB/s
FP32_MATRIX_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 / 32x32x2 (f32 in, f32 accumulate), whole chip (MI355X_MICROARCH.md)
F16_MATRIX_PEAK_TFLOPS = 2500.0  # v_mfma_f32_16x16x32_f16 / _bf16, dense (MI355X_MICROARCH.md: ~2.5 PF; never the 2:1-sparsity figure)
PLAIN_SPMM = ('stc_patch_spmm_f32', 'stc_patch_spmm_bf16', 'stc_bcsr_spmm_f32', 'stc_bcsr_spmm_bf16')
# every C-ABI entry point that is an aggregation Y = S.X (+ epilogue)
SPMM_ENTRY_POINTS = ('stc_patch_spmm_f32', 'stc_patch_spmm_bf16', 'stc_bcsr_spmm_f32', 'stc_csr_spmm_f32', 'stc_spmm_sum_f32', 'stc_spmm_blend_fwd_f32',
                     'stc_bcsr_spmm_bf16', 'stc_csr_spmm_bf16', 'stc_spmm_sum_bf16', 'stc_spmm_blend_fwd_bf16',
                     'stc_ring2_sum_f32', 'stc_ring2_blend_f32', 'stc_ring2_chain_f32')
# what the timed region records HIP events for: the roofline kernel (plain aggregation) and the entry points that can dominate a step
PRICED_ENTRY_POINTS = PLAIN_SPMM + ('stc_cell_bwd_planar_f32', 'stc_ring2_sum_f32', 'stc_ring2_blend_f32', 'stc_ring2_chain_f32', 'stc_cell_small_fwd_f32', 'stc_cell_small_bwd_f32')
METRIC = 'STC-GNN fwd+bwd samples/sec at N=50k,C=32; SpMM HBM GB/s vs peak, 1/8 GPU'
CPU_WARMUP, CPU_TIMED = 2, 7    # 2 warm-up + 7 timed iterations, median


def parse(argv=None):
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
    ap.add_argument('--batch-per-gpu', type=int, default=8,
                    help='samples per GPU (weak scaling); ~30 GB of saved activations per sample: 8 = 243 GB of the 309 GB a MI355X reports.  '
                         'Same box, same library: 31.4 samples/s at 5, 31.6 at 6, 32.1 at 8 (longer launches: shorter tails, fewer boundaries per sample)')
    ap.add_argument('--global-batch', type=int, default=0, help='total batch over all GPUs (strong scaling: each rank takes global-batch / gpus samples); 0 = weak scaling')
    ap.add_argument('--preset', choices=('cfg2', 'cfg4', 'cfg5', 'sf', 'sf-learned'), default=None,
                    help="BASELINE.json's other configurations through the same bench (not the metric): cfg4 = N 10 000, K = 3, batch 16; cfg5 = C = 64, "
                         'bf16 state storage; sf = the SF-incidents shape (N = 100, C = 5, T = 9 + 3, batch 32, fixed sparse graph); sf-learned = the same shape with '
                         "the reference's FULL model: MGP_Gen's learned dense graphs (2e8 parameters in MixedFusion), fused Adam; cfg2 = one BDG_Dif layer "
                         '(B = 32, N = 200, C = 8, L = 32, Ho = 32), forward and forward + backward')
    ap.add_argument('--graph-mode', choices=('csr-fixed', 'dense-learned'), default='csr-fixed',
                    help="csr-fixed = the metric's mode (caller-supplied sparse Gs); dense-learned = the reference's semantics (MGP_Gen, N <~ 300)")
    ap.add_argument('--permute', action='store_true', help='random node order (seed 1234) instead of row-major')
    ap.add_argument('--no-reorder', action='store_true', help='keep the given node order (skip the internal RCM renumbering)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-shots', default=f'{CPU_WARMUP},{CPU_TIMED}', help='warm-up,timed shots of the CPU baseline (SURVEY 8(d4): 2,7)')
    ap.add_argument('--no-unit-d3', action='store_true', help='skip the cold SURVEY 8(d3) SpMM unit measurement')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the side measurements of the default line (alt_formats.bf16x3 and permuted: 1 + 3 steps each, same process)')
    ap.add_argument('--no-presets', action='store_true',
                    help="skip the `presets` block of the default line (one GPU): BASELINE.json's other configurations, 3 timed steps each, each in a child process after the metric's own measurements")
    ap.add_argument('--hip-graph', action='store_true',
                    help='(one GPU; not the default line) time replays of the train step captured into a HIP graph: removes the launch gaps that '
                         'matter at small N; the per-kernel events then come from eager steps after the timed region')
    ap.add_argument('--eager', dest='hip_graph', action='store_false', help='per-launch dispatch where a preset defaults to HIP-graph replay (--preset sf)')
    ap.add_argument('--storage', choices=('f32', 'bf16'), default='f32',
                    help='storage type of states / gates / their gradients: f32 = the metric (reference arithmetic); bf16 = BASELINE configuration 5 '
                         '(not the metric: bf16 has no reference behaviour), fp32 parameters and fp32 sums inside every kernel')
    # a preset changes DEFAULTS (explicit flags still win): found first, then the full parse
    pre, _ = ap.parse_known_args(argv)
    if pre.preset == 'cfg4':
        # ~3 000 launches of 5 - 25 us per step: replayed from one captured HIP graph on one GPU (63.1 -> 65.2 samples/s; --eager: per-launch dispatch)
        # 16 samples per GPU (169 GB): same box, 4 / 8 / 12 / 16 / 24 samples: 71.4 / 76.2 / 77.1 / 77.9 / 77.7 samples/s
        ap.set_defaults(grid=100, order=3, batch_per_gpu=16, hip_graph=True)
    elif pre.preset == 'cfg5':
        # (a C = 64 oracle cell is ~20 s on the host: fewer shots; 5 samples per GPU: at 8 -- 261 GB -- it measured 26.5 against 26.6 samples/s)
        ap.set_defaults(categories=64, storage='bf16', cpu_shots='1,3', batch_per_gpu=5)
    elif pre.preset == 'cfg2':
        # one layer = a handful of 5 - 30 us launches: the host's dispatch, not the chip, sets the eager time; `value` is over HIP-graph replays
        # of the captured forward + backward (--eager: per-launch dispatch; both are in the line)
        ap.set_defaults(hip_graph=True)
    elif pre.preset == 'sf':
        # launch-bound (~700 launches of ~7 us per step): the step is replayed from ONE captured HIP graph (--eager keeps per-launch dispatch)
        ap.set_defaults(grid=10, categories=5, obs=9, pred=3, batch_per_gpu=32, no_unit_d3=True, hip_graph=True)
    elif pre.preset == 'sf-learned':
        ap.set_defaults(grid=10, categories=5, obs=9, pred=3, batch_per_gpu=32, no_unit_d3=True, hip_graph=True, graph_mode='dense-learned')
    a = ap.parse_args(argv)
    if a.global_batch:
        if a.global_batch % a.gpus:
            ap.error(f'--global-batch {a.global_batch} is not divisible by --gpus {a.gpus}: shards must be equal')
        a.batch_per_gpu = a.global_batch // a.gpus
    return a


def self_launch(a) -> int:
    """``python bench.py --gpus N`` outside torchrun: start the N ranks as a CHILD process tree and wait for it.

    Runs before this process has imported torch or touched the GPU; the child is ``python -m torch.distributed.run``
    with the driver's own argument form (one node, 127.0.0.1 rendezvous).  Output is inherited, so rank 0's JSON line is
    this command's JSON line; the exit code is the child's."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // max(1, a.gpus))))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


def cpu_baseline_small(a, Gs_dense, Gc, sd_cpu, As_dense=None):
    """Small graphs (the SF shape and its relatives): the oracle's WHOLE model -- the reference's algorithm op for op, dense einsum included
    (oracle.encdec_forward; with learned graphs oracle.stcgnn_forward, MGP_Gen and its 2 N^4 MixedFusion parameters included) -- forward,
    ComboLoss and backward at the bench's own batch on the host cores; no optimizer step (the GPU value includes Adam)."""
    import torch
    from oracle import stc_oracle as O
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    warm, timed = (int(v) for v in a.cpu_shots.split(','))
    N, C, B = a.grid * a.grid, a.categories, a.batch_per_gpu
    g = torch.Generator().manual_seed(1000)
    X = (torch.rand(B, a.obs, N, C, generator=g) < 0.1635).float()
    Y = (torch.rand(B, a.pred, N, C, generator=g) < 0.1635).float()
    leaves = {k: v.clone().requires_grad_() for k, v in sd_cpu.items()}
    ts = []
    for shot in range(warm + timed):
        for v in leaves.values():
            v.grad = None
        t0 = time.perf_counter()
        if As_dense is not None:
            yhat = O.stcgnn_forward(X, As_dense, Gc, leaves, a.order, a.order, a.hidden, a.layers, a.pred)
        else:
            yhat = O.encdec_forward(X, Gs_dense, Gc, leaves, a.order, a.order, a.hidden, a.layers, a.pred)
        O.combo_loss(yhat, Y).backward()
        ts.append(time.perf_counter() - t0)
    t = _median(ts[warm:])
    return dict(value=B / t, unit='samples/s', cores=threads, kind='port',
                sample=f'the oracle\'s whole model (reference algorithm op for op: dense einsum, matrix-side cheby_poly'
                       f'{", MGP_Gen + MixedFusion" if As_dense is not None else ""}) forward + ComboLoss + backward, batch {B}, no optimizer step; '
                       f'{warm} warm-up + {timed} timed, median {t:.3f} s per step',
                shots_s=[round(v, 4) for v in ts])


def cpu_baseline(a, GsT_sparse, Gc, sd_cpu):
    """Oracle fwd+bwd on the host cores for a bounded sample of the same workload: 2 warm-up + 7 timed, median.

    Sample: one STC_Cell of each kind the model has (layer 0: 1 + 16 input columns; the others: 16 + 16), batch 1, same
    graph / width / parameters, forward + backward through ``oracle.stc_cell``; a full sample is ``layers * (obs + pred)``
    such cells, so samples/s = 1 / (n0 t0 + n1 t1).  kind = "port": the oracle's sparse feature-side restatement (the
    reference itself cannot run N = 50 176: dense Gs + eye = 20 GB and an N^3 cheby_poly), validated against the dense
    reference at N = 1 024 / 10 000 in tests/.  ``dense_anchor`` ties the port to the true reference: the oracle's DENSE
    ``bdg_dif`` (the reference's algorithm op for op) at BASELINE.md section 2's N = 10 000 shape (0.90 s there, 8 cores).
    """
    import torch
    from oracle import stc_oracle as O
    threads = min(os.cpu_count() or 1, 32)      # more threads than this only slow the torch CPU ops down on a 2-socket host
    torch.set_num_threads(threads)
    CPU_WARMUP, CPU_TIMED = (int(v) for v in a.cpu_shots.split(','))
    N, C, h, K = a.grid * a.grid, a.categories, a.hidden, a.order
    g = torch.Generator().manual_seed(0)

    def one_cell(prefix, cin):
        Xt = (torch.rand(1, N, C, cin, generator=g) < 0.1635).float() if cin == 1 else torch.rand(1, N, C, cin, generator=g) - 0.5
        Xt.requires_grad_(cin != 1)
        H = (torch.rand(1, N, C, h, generator=g) - 0.5).requires_grad_()
        R = torch.rand(1, N, C, h, generator=g)
        p = [sd_cpu[f'{prefix}.{n}'].clone().requires_grad_() for n in ('gates.W', 'gates.b', 'candi.W', 'candi.b')]
        ts = []
        for shot in range(CPU_WARMUP + CPU_TIMED):
            t0 = time.perf_counter()
            out = O.stc_cell(GsT_sparse, Gc, Xt, H, *p, K, K, conv=O.bdg_dif_sparse)
            (out * R).sum().backward()
            ts.append(time.perf_counter() - t0)
        return _median(ts[CPU_WARMUP:]), ts

    t0, shots0 = one_cell('encoder.cell_list.0', 1)
    t1, shots1 = one_cell('encoder.cell_list.1' if a.layers > 1 else 'decoder.cell_list.0', h)
    n0 = a.obs                                                  # encoder layer 0 is the only cell with a 1-column input
    n1 = a.layers * (a.obs + a.pred) - n0
    per_sample = n0 * t0 + n1 * t1
    out = dict(value=1.0 / per_sample, unit='samples/s', cores=threads, kind='port',
               sample=f'oracle.stc_cell (torch CPU, sparse feature-side variant) fwd+bwd, batch 1, same graph/width/parameters: one layer-0 cell '
                      f'(median {t0:.2f} s) and one 16+16 cell (median {t1:.2f} s), {CPU_WARMUP} warm-up + {CPU_TIMED} timed each; a sample = {n0} + {n1} such cells '
                      f'= {per_sample:.0f} s',
               shots_s=dict(layer0=[round(t, 3) for t in shots0], wide=[round(t, 3) for t in shots1]))
    if a.preset == 'cfg4' and N <= 10000 and not os.environ.get('STC_BENCH_NO_DENSE_ANCHOR'):
        # configuration 4's own anchor: ONE STC_Cell forward of the reference's dense algorithm at this N and order (7.9 s on 8
        # cores, most of it the two N^3 matrix products of cheby_poly), 1 warm-up + 3 timed
        try:
            gd = torch.Generator().manual_seed(5)
            Gs = torch.rand(N, N, generator=gd)
            Gs /= Gs.sum(-1, keepdim=True)
            Xt = (torch.rand(1, N, C, 1, generator=gd) < 0.1635).float()
            Ht = torch.rand(1, N, C, h, generator=gd) - 0.5
            p = [sd_cpu[f'encoder.cell_list.0.{n}'] for n in ('gates.W', 'gates.b', 'candi.W', 'candi.b')]
            ts = []
            with torch.no_grad():
                for shot in range(4):
                    t = time.perf_counter()
                    O.stc_cell(Gs, Gc, Xt, Ht, *p, K, K)
                    ts.append(time.perf_counter() - t)
            out['dense_cell_anchor'] = dict(what=f'oracle.stc_cell forward (the reference algorithm: dense einsum, matrix-side cheby_poly of order {K}), B=1 N={N} C={C} '
                                                 'hidden 16, dense Gs; 1 warm-up + 3 timed, median', seconds=_median(ts[1:]), cores=threads,
                                            reference_in_survey_container_s=7.9, reference_cores=8)
            del Gs
        except MemoryError:
            pass
    if not os.environ.get('STC_BENCH_NO_DENSE_ANCHOR'):
        try:
            Nd = 10000
            gd = torch.Generator().manual_seed(3)
            Gs = torch.rand(Nd, Nd, generator=gd)
            Gs /= Gs.sum(-1, keepdim=True)
            X = torch.rand(1, Nd, 32, 32, generator=gd)
            W = torch.randn(4 * 32, 32, generator=gd) * 0.1
            Gc32 = torch.softmax(torch.randn(32, 32, generator=gd), -1)
            ts = []
            with torch.no_grad():
                for shot in range(4):
                    t = time.perf_counter()
                    O.bdg_dif(X, Gs, Gc32, W, None, 2, 2)
                    ts.append(time.perf_counter() - t)
            out['dense_anchor'] = dict(what='oracle.bdg_dif forward (the reference algorithm: dense einsum, matrix-side cheby_poly, concat + projection), '
                                            'B=1 N=10000 C=32 L=32 Ho=32 K=2, dense 400 MB Gs; 1 warm-up + 3 timed, median',
                                       seconds=_median(ts[1:]), cores=threads,
                                       reference_in_survey_container_s=0.90, reference_cores=8)
        except MemoryError:
            pass
    return out


def spmm_unit_d3(graph, dev, C, L, dtype, launches=60, rotate=6, reorder=True):
    """SURVEY 8(d3)'s unit of work exactly: one application of the FORWARD operand, Y = S.X with S = Gs^T (reference STC_GNN.py:37),
    over rows of F = C*L values, B = 1, timed COLD (operands rotated over ``rotate`` buffer pairs >> the 256 MiB Infinity Cache)
    with HIP events on the launch stream.  The graph is taken in the node order the timed step uses: the internally renumbered
    copy (``with_locality``) when the model renumbers, the given order otherwise."""
    import torch
    from stc_hip import ops
    from stc_hip.graph import csr_operand
    k = ops.kernels()
    renumbered = False
    if reorder:
        g2, order = graph.with_locality()
        renumbered, graph = order is not None, g2
    op = csr_operand(graph, dev)
    N, F = graph.n, C * L
    Xs = [torch.randn(1, N, F, device=dev).to(dtype) for _ in range(rotate)]
    Ys = [torch.empty(1, N, F, device=dev, dtype=dtype) for _ in range(rotate)]

    def run(i):
        k.csr_spmm(op.fwd_rowptr, op.fwd_colidx, op.fwd_val, N, N, Xs[i % rotate], None, Ys[i % rotate], 1.0, 0.0, plan=op.fwd_plan)

    for i in range(rotate):
        run(i)
    torch.cuda.synchronize(dev)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stream = torch.cuda.current_stream(dev)
    s.record(stream)
    for i in range(launches):
        run(i)
    e.record(stream)
    torch.cuda.synchronize(dev)
    us = 1e3 * s.elapsed_time(e) / launches
    nbytes = graph.nnz * 8 + 4 * (N + 1) + 2 * N * F * Xs[0].element_size()
    return dict(what=f'Y = S.X (S = Gs^T, the forward operand), B=1, F=C*L={F}, N={N}, nnz={graph.nnz}, node order '
                     f'{"renumbered internally (as the timed step)" if renumbered else "as given (as the timed step)"}, {launches} cold launches '
                     f'(operands rotated over {rotate} x {2 * N * F * Xs[0].element_size() / 1e6:.0f} MB); the binding\'s default dispatch: one sample of '
                     f'rows this wide is below the patch kernel\'s launch-size rule, i.e. the row-blocked kernel (stc_bcsr_spmm_*)',
                algorithmic_bytes=nbytes, avg_launch_us=us, achieved=nbytes / us / 1e3, unit='GB/s', frac=nbytes / us / 1e3 / HBM_PEAK_GBPS,
                target_frac=0.40)


def csrc_sha():
    """Hash of the kernel sources: a PMC traffic file is only quoted by the bench line while it matches."""
    import hashlib
    d = os.path.join(REPO, 'stc-gnn_amd', 'csrc')
    hsh = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            with open(os.path.join(d, f), 'rb') as fh:
                hsh.update(f.encode() + b'\0' + fh.read())
    return hsh.hexdigest()[:16]


def _profile_doc(name):
    """The newest ``profiles/r*/<name>`` (highest round first): PMC files are committed per round."""
    root = os.path.join(REPO, 'profiles')
    for rnd in sorted((d for d in os.listdir(root) if d.startswith('r') and d[1:].isdigit()), key=lambda d: -int(d[1:])) if os.path.isdir(root) else ():
        path = os.path.join(root, rnd, name)
        if os.path.exists(path):
            with open(path) as fh:
                return json.load(fh), f'profiles/{rnd}/{name}'
    return None, None


def pmc_traffic(a, config_key):
    """HBM bytes per plain SpMM launch from a PMC pass over THIS command (tools/gpu_pmc_bench.sh: FETCH_SIZE and WRITE_SIZE
    in separate ``rocprofv3 --pmc`` runs), quoted only if the file was collected on the same kernel sources (csrc_sha)
    and the same configuration; otherwise null.  Never a number from an older build."""
    doc, where = _profile_doc('hbm_traffic_bench.json')
    if doc is None:
        return None, None
    if doc.get('csrc_sha') != csrc_sha() or doc.get('config_key') != config_key:
        return None, f'{where} is from other sources/configuration ({doc.get("csrc_sha")}, {doc.get("config_key")}): not quoted'
    import re
    # the plain launches only: the patch kernel without Y0, or the row-blocked kernel with MODE = 0 (EP_PLAIN)
    ks = [v for name, v in doc['kernels'].items() if re.match(r'spmm_patch_kernel<\d+, false|spmm_bcsr_kernel<\d+, 0,', name)]
    if not ks:
        return None, None
    traffic = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in ks) / sum(v['launches'] for v in ks)
    return traffic, ('PMC (FETCH_SIZE, WRITE_SIZE in separate rocprofv3 --pmc passes, unit-corrected as MI355X_MICROARCH.md prescribes), mean over the '
                     f'plain aggregation launches of one step of this command on these sources (csrc {doc["csrc_sha"]}): {where}')


# C-ABI entry point -> the kernel it launches, as the PMC file names it
PMC_KERNEL_OF = {'stc_cell_bwd_planar_f32': r'cell_bwd_x3_kernel<', 'stc_cell_gates_fwd_planar_f32': r'node_fwd_x3_kernel<',
                 'stc_ring2_sum_f32': r'ring2_sum_kernel<0,', 'stc_ring2_blend_f32': r'ring2_sum_kernel<1,', 'stc_ring2_chain_f32': r'ring2_sum_kernel<2,'}


def pmc_entry_traffic(entry_point, config_key):
    """Mean HBM bytes per launch of one entry point's kernel from the same committed PMC file as ``pmc_traffic`` (same validity rule), or None."""
    doc, _ = _profile_doc('hbm_traffic_bench.json')
    pat = PMC_KERNEL_OF.get(entry_point)
    if doc is None or pat is None or doc.get('csrc_sha') != csrc_sha() or doc.get('config_key') != config_key:
        return None
    import re
    ks = [v for name, v in doc['kernels'].items() if re.match(pat, name)]
    return sum(v['hbm_bytes_per_launch'] * v['launches'] for v in ks) / sum(v['launches'] for v in ks) if ks else None


def pmc_mfma(config_key):
    """Matrix-pipe / vector-pipe busy fractions of the projection kernels (the split-operand MFMA cell kernels) from the committed
    ``rocprofv3 --pmc`` passes over THIS command (tools/gpu_pmc_mfma.sh -> profiles/rNN/mfma_util.json): quoted only while the
    file's kernel-source hash and configuration equal the running tree's, else null with a note."""
    doc, where = _profile_doc('mfma_util.json')
    if doc is None:
        return None
    if doc.get('csrc_sha') != csrc_sha() or doc.get('config_key') != config_key:
        return {'value': None, 'note': f'{where} is from other sources/configuration ({doc.get("csrc_sha")}, {doc.get("config_key")}): not quoted'}
    keep = {name: {'launches': v['launches'], 'mfma_busy': v['mfma_util'], 'valu_busy': v['valu_busy']} for name, v in doc['kernels'].items()
            if 'cell_bwd' in name or 'node_fwd' in name or 'node_bwd' in name}
    return {'source': 'committed', 'file': where, 'csrc_sha': doc['csrc_sha'],
            'definition': doc.get('definition'),
            'peak_note': 'busy fraction of the matrix pipe while the kernel runs; 1.0 = the dense MFMA peak of the instruction in use',
            'kernels': keep}


def projection_rates(a, per_kernel, R, C):
    """Algorithmic flops of the dense projections + category mixes (2 B N C K^2 L Ho per projection) of the planar cell kernels of
    the metric step, over the measured time of those launches, against the dense f16 MFMA peak -- the live counterpart of the PMC busy
    fractions.  The kernels are HBM-bound (one / two waves per SIMD): the matrix pipe is far from its peak by design; this says how far.
    ``matrix_instructions`` = what the pipe really executes: three fp16 piece products per fp32 product (six bf16 on the bf16 x 3 format)."""
    fwd, bwd = per_kernel.get('stc_cell_gates_fwd_planar_f32'), per_kernel.get('stc_cell_bwd_planar_f32')
    if not fwd or not bwd or a.order != 2 or a.hidden != 16:
        return None
    h, K = a.hidden, a.order
    wide_cells = (a.layers - 1) * a.obs + a.layers * a.pred
    narrow_cells = a.obs

    def cell(Lw):                                                     # forward: gates (2h) + candidate (h) projections, their category mixes
        proj = 2.0 * R * C * (K * K * Lw) * 3 * h
        mix = 2.0 * R * C * C * (K - 1) * (2 * h + K * h)
        return proj + mix
    f_fwd = wide_cells * cell(2 * h) + narrow_cells * cell(1 + h)
    f_bwd = 2.0 * f_fwd                                               # dZ and dW: each the forward's products once more
    out = {'what': 'algorithmic flops (projections + category mixes, fp32 products) of the planar cell launches of one step / their measured time',
           'peak': F16_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'pieces_per_product': 3}
    for name, d, fl in (('gates_forward', fwd, f_fwd), ('cell_backward', bwd, f_bwd)):
        per_step_s = d['ms'] / a.steps / 1e3
        tf = fl / per_step_s / 1e12 if per_step_s > 0 else 0.0
        out[name] = {'flops_per_step': fl, 'achieved': tf, 'frac': tf / F16_MATRIX_PEAK_TFLOPS,
                     'matrix_instructions': {'achieved': 3 * tf, 'frac': 3 * tf / F16_MATRIX_PEAK_TFLOPS}}
    return out


class PowerSampler:
    """Package power and shader clock of this rank's GPU, sampled ~10 times a second over the timed region (amdsmi; rank 0 reports).
    The metric step runs at the package power limit with the shader clock throttled (profiles/r03/clock_trace.txt), so the line says
    under which power / clock its value was measured.  Never allowed to break the line: any failure -> ``"power": null``."""

    def __init__(self, index):
        import threading
        self.rows, self.stop_flag, self.thread, self.err = [], threading.Event(), None, None
        try:
            import amdsmi
            import torch
            self.smi = amdsmi
            amdsmi.amdsmi_init()
            handles = amdsmi.amdsmi_get_processor_handles()
            # amdsmi lists every physical GPU of the host, whatever HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES say and in its own order:
            # the rank's device is found by its PCI bus id, not by its HIP index (no match: no power block rather than another GPU's)
            want = self._bdf(torch.cuda.get_device_properties(index).pci_bus_id) if hasattr(torch.cuda.get_device_properties(index), 'pci_bus_id') else None
            self.h = None
            for h_ in handles:
                if want is not None and self._bdf(amdsmi.amdsmi_get_gpu_device_bdf(h_)) == want:
                    self.h = h_
            if self.h is None:
                if want is None and len(handles) == 1:
                    self.h = handles[0]
                else:
                    raise RuntimeError(f'no amdsmi handle with PCI bus id {want}')
            self.limit = self._num(amdsmi.amdsmi_get_power_info(self.h).get('power_limit'))
            self.rated = self._num(amdsmi.amdsmi_get_clock_info(self.h, amdsmi.AmdSmiClkType.GFX).get('max_clk'))
            self.thread = threading.Thread(target=self._run, daemon=True)
        except Exception as e:                                       # noqa: BLE001 -- no amdsmi, no permission, no such device: no power block
            self.err = repr(e)

    @staticmethod
    def _num(v):
        return float(v) if isinstance(v, (int, float)) else None

    @staticmethod
    def _bdf(v):
        """'0000:c3:00.0' / 'c3:00.0' / 195 (a bare bus number) -> (bus, device, function), None if unreadable."""
        try:
            if isinstance(v, int):
                return (v, 0, 0)
            parts = str(v).lower().replace('.', ':').split(':')
            bus, dev_, fn = parts[-3], parts[-2], parts[-1]
            return (int(bus, 16), int(dev_, 16), int(fn, 16))
        except Exception:                                            # noqa: BLE001
            return None

    def _run(self):
        while not self.stop_flag.is_set():
            try:
                w = self._num(self.smi.amdsmi_get_power_info(self.h).get('current_socket_power'))
                c = self._num(self.smi.amdsmi_get_clock_info(self.h, self.smi.AmdSmiClkType.GFX).get('clk'))
                if w is not None and c is not None:
                    self.rows.append((w, c))
            except Exception as e:                                   # noqa: BLE001
                self.err = repr(e)
                return
            self.stop_flag.wait(0.1)

    def start(self):
        if self.thread is not None:
            self.thread.start()

    def stop(self):
        if self.thread is None:
            return None
        self.stop_flag.set()
        self.thread.join(timeout=2.0)
        if not self.thread.is_alive():                               # (still inside an amdsmi call after the timeout: leave the library up)
            try:
                self.smi.amdsmi_shut_down()
            except Exception:                                        # noqa: BLE001
                pass
        if not self.rows:
            return None
        w, c = [r[0] for r in self.rows], [r[1] for r in self.rows]
        limit_w = self.limit / 1e6 if self.limit and self.limit > 1e5 else self.limit       # (amdsmi reports the limit in microwatts)
        return {'source': 'amdsmi, ~10 samples/s over the timed region', 'samples': len(w), 'package_w_mean': sum(w) / len(w), 'package_w_max': max(w),
                'package_w_limit': limit_w, 'sclk_mhz_mean': sum(c) / len(c), 'sclk_mhz_min': min(c), 'sclk_mhz_rated': self.rated}


PRESET_RUNS = (('cfg2', ['--preset', 'cfg2', '--order', '2']), ('cfg2-order3', ['--preset', 'cfg2', '--order', '3']), ('cfg4', ['--preset', 'cfg4']),
               ('cfg5', ['--preset', 'cfg5']), ('sf', ['--preset', 'sf']), ('sf-learned', ['--preset', 'sf-learned']))


def run_presets(parent_reserved_gb, steps=3, timeout_s=150):
    """BASELINE.json's other configurations for the default line: each ``--preset`` as a CHILD process of this bench (3 timed steps, 1 warm-up, HIP-graph
    replays where the preset defaults to them, no CPU baseline), one after the other, after the metric's own measurements; the digest of each child's line.
    Never re-execs this process (it has touched the GPU); a child that fails or overruns leaves an ``error`` entry instead of ending the line."""
    digest = {'what': f'python bench.py --preset <name> --steps {steps} --warmup 1 --no-cpu-baseline --no-unit-d3, one child process each, run after the timed region '
                      'of the metric (its memory handed back first); not the metric -- the full lines of builder runs are profiles/rNN/preset_*.json',
              'parent_reserved_gb_while_they_ran': parent_reserved_gb}
    if parent_reserved_gb > 60.0:
        digest['error'] = 'the metric run still holds its memory: presets skipped'
        return digest
    env = dict(os.environ)
    for key in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(key, None)
    only = [v for v in os.environ.get('STC_BENCH_PRESETS', '').split(',') if v]         # (tests: a subset by name)
    for name, flags in PRESET_RUNS:
        if only and name not in only:
            continue
        t0 = time.perf_counter()
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(steps), '--warmup', '1', '--no-cpu-baseline',
                                  '--no-unit-d3'] + flags, env=env, capture_output=True, text=True, timeout=timeout_s)
            lines = [ln for ln in res.stdout.strip().split('\n') if ln.startswith('{')]
            if res.returncode != 0 or not lines:
                digest[name] = {'error': f'exit code {res.returncode}: ' + (res.stderr.strip().split('\n') or [''])[-1][:300]}
                continue
            d = json.loads(lines[-1])
            roof = d.get('roofline') or {}
            dom = roof.get('dominant') or {}
            digest[name] = {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'], 'hip_graph': d.get('hip_graph'),
                            'dtype': d.get('dtype'), 'workload': d['config']['workload'],
                            'roofline': {'bound': roof.get('bound'), 'kernel': (roof.get('kernel') or '')[:160], 'achieved': roof.get('achieved'),
                                         'peak': roof.get('peak'), 'unit': roof.get('unit'), 'frac': roof.get('frac'),
                                         'dominant': {k: dom.get(k) for k in ('entry_point', 'share_of_kernel_time', 'avg_launch_us', 'achieved', 'frac') if k in dom}},
                            **({'forward_ms': d['forward_ms']} if 'forward_ms' in d else {}), **({'eager': d['eager']} if 'eager' in d else {}),
                            'child_run_s': time.perf_counter() - t0}
        except subprocess.TimeoutExpired:
            digest[name] = {'error': f'no line within {timeout_s} s'}
        except Exception as e:                                       # a malformed child line must not cost the metric's own line
            digest[name] = {'error': f'{type(e).__name__}: {e}'[:300]}
    return digest


def main():
    t_start = time.perf_counter()
    a = parse()
    if a.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(self_launch(a))            # before torch / the GPU are touched: the ranks are children of this process
    if a.preset == 'cfg2':
        return bench_cfg2(a)

    for _p in (REPO, os.path.join(REPO, 'stc-gnn_amd')):
        if _p not in sys.path:
            sys.path.insert(0, _p)
    import torch
    import torch.distributed as dist
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
    learned = a.graph_mode == 'dense-learned'
    model = M.STCGNN(N, C, a.order, a.order, 1, a.hidden, a.layers, a.pred, graph_mode=a.graph_mode, reorder_nodes=not a.no_reorder,
                     storage_dtype=torch.bfloat16 if a.storage == 'bf16' else torch.float32)
    sd_cpu = None if a.no_cpu_baseline else {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    Gc = Gc_cpu.to(dev)
    # learned graphs: MGP_Gen takes the raw adjacencies (the reference's Main.py hands it the 0/1 grid adjacency and a category matrix)
    As_in = CsrGraph.queen_grid(a.grid, a.grid, normalize=False).to_dense().to(dev) if learned else graph
    # SURVEY 8(d1) names seeds 0 / 1 for inputs / targets; a sharded batch needs a different draw per rank, so rank r uses
    # 1000 + r for both (rank 0 of a 1-GPU run: seed 1000).  Bernoulli(0.1635) either way.
    g = torch.Generator().manual_seed(1000 + rank)
    X = (torch.rand(B, a.obs, N, C, generator=g) < 0.1635).float().to(dev)
    Y = (torch.rand(B, a.pred, N, C, generator=g) < 0.1635).float().to(dev)
    crit = ComboLoss()
    bucket = sdist.GradBucket(model.parameters())
    graphed = a.hip_graph and world == 1
    # learned graphs: the two MixedFusion matrices (2 x 400 MB at the SF shape) on stc_adam_f32, the rest on torch's fused Adam (stc_hip/optim.py)
    from stc_hip.optim import Adam as StcAdam
    opt = (StcAdam if learned else torch.optim.Adam)(model.parameters(), lr=2e-3, weight_decay=1e-4, capturable=graphed, **({'fused': True} if learned else {}))
    stream = torch.cuda.current_stream(dev)
    marks = []                                                                  # (t_begin, t_backward_done, t_allreduce_done, t_adam_done) events

    def step(mark=False):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if mark else None
        if mark:
            ev[0].record(stream)
        bucket.zero()
        loss = crit(model(X_seq=X, As=As_in, Ac=Gc), Y)
        loss.backward()
        if mark:
            ev[1].record(stream)
        bucket.allreduce_mean()
        if mark:
            ev[2].record(stream)
        opt.step()
        if mark:
            ev[3].record(stream)
            marks.append(ev)
        return loss

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()
    t_built = time.perf_counter()                                                # process start -> graph, model, inputs resident
    for _ in range(max(a.warmup, 2 if graphed else 0)):
        step()
    hip = ops.kernels()
    fence()
    t_warm = time.perf_counter()                                                 # ... -> warm-up done (first launches: graph renumbering, plans, workspaces)
    if graphed:
        # the whole step as ONE captured HIP graph: `value` is over replays; the launch timer cannot run inside a replay, so the per-kernel
        # events (roofline, kernels) are taken from eager steps after the timed region
        torch.cuda.empty_cache()                                     # the graph gets a memory pool of its own: hand the eager pool back first
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            step()
        cg.replay()
        fence()
        sampler = PowerSampler(local)
        sampler.start()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            cg.replay()
        fence()
        elapsed = time.perf_counter() - t0
        power = sampler.stop()
        del cg
        torch.cuda.empty_cache()
        hip.timer = KernelTimer()
        for _ in range(min(a.steps, 3)):
            loss = step(mark=True)
        fence()
        per_kernel = hip.timer.summary()
        for d_ in per_kernel.values():                               # scale the event sums to the step count of the line
            for key in ('launches', 'ms', 'bytes'):
                d_[key] = d_[key] * a.steps / min(a.steps, 3)
                for t_ in d_.get('tags', {}).values():
                    t_[key] = t_[key] * a.steps / min(a.steps, 3)
        hip.timer = None
    else:
        # timed region: HIP events around the launches the roofline block prices (the plain aggregation and the dominant cell kernels), not
        # around all ~700 launches of a step -- 1 400 event records per step cost 2 ms of the 174; the complete per-kernel table comes
        # from two further steps with every launch timed, scaled to the step count (the priced entry points keep their timed-region numbers)
        hip.timer = KernelTimer(only=PRICED_ENTRY_POINTS)
        sampler = PowerSampler(local)
        sampler.start()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step(mark=True)
        fence()
        elapsed = time.perf_counter() - t0
        power = sampler.stop()
        priced = hip.timer.summary()
        hip.timer = KernelTimer()
        table_steps = min(a.steps, 2)
        for _ in range(table_steps):
            step()
        fence()
        per_kernel = hip.timer.summary()
        hip.timer = None
        for d_ in per_kernel.values():
            for key in ('launches', 'ms', 'bytes'):
                d_[key] = d_[key] * a.steps / table_steps
                for t_ in d_.get('tags', {}).values():
                    t_[key] = t_[key] * a.steps / table_steps
        per_kernel.update(priced)
    n_ranks_seen = 1
    per_rank = {'ms_per_step': [1e3 * elapsed / a.steps], 'build_s': [t_built - t_start], 'warmup_s': [t_warm - t_built]}
    if dist.is_initialized():
        mine = torch.tensor([elapsed, t_built - t_start, t_warm - t_built], device=dev, dtype=torch.float64)
        every_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every_rank, mine)
        per_rank = {'ms_per_step': [1e3 * float(t[0]) / a.steps for t in every_rank], 'build_s': [float(t[1]) for t in every_rank],
                    'warmup_s': [float(t[2]) for t in every_rank]}
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                                   # every rank really took part in a collective
        n_ranks_seen = int(ones.item())
    phases = [sum(ev[i].elapsed_time(ev[i + 1]) for ev in marks) / max(1, len(marks)) for i in range(3)]

    # the same step with the per-launch event records off: what the instrumentation of the timed region costs
    untimed_steps = min(a.steps, 3)
    fence()
    t1 = time.perf_counter()
    for _ in range(untimed_steps):
        step()
    fence()
    untimed_ms = 1e3 * (time.perf_counter() - t1) / max(1, untimed_steps)

    # Side measurements of the metric line (one GPU, default configuration): what used to exist as builder-run numbers only.
    #   alt_formats.bf16x3   the same step on the 24-bit operand format (three bf16 pieces, six products; the default keeps 22 bits)
    #   permuted             SURVEY 8(d1) variant B: the same graph under a seeded random node order (the host renumbers it: reverse
    #                        Cuthill-McKee, then greedy clusters for the patch form), with the rate of its plain aggregation launches
    extras = {}
    if rank == 0 and world == 1 and not graphed and not a.no_extras and a.preset is None and not learned and a.storage == 'f32':
        from stc_hip import _lib as _l

        def side(model_, As_, n=3):
            bucket_ = bucket if model_ is model else sdist.GradBucket(model_.parameters())
            opt_ = opt if model_ is model else torch.optim.Adam(model_.parameters(), lr=2e-3, weight_decay=1e-4)

            def st():
                bucket_.zero()
                crit(model_(X_seq=X, As=As_, Ac=Gc), Y).backward()
                bucket_.allreduce_mean()
                opt_.step()
            t_a = time.perf_counter()
            st()
            fence()
            first_s = time.perf_counter() - t_a
            hip.timer = KernelTimer(only=PLAIN_SPMM)
            t_a = time.perf_counter()
            for _ in range(n):
                st()
            fence()
            dt = time.perf_counter() - t_a
            got = hip.timer.summary()
            hip.timer = None
            pl = dict(launches=0, ms=0.0, bytes=0)
            for name in PLAIN_SPMM:
                tv = got.get(name, {}).get('tags', {}).get('plain')
                if tv:
                    for key in pl:
                        pl[key] += tv[key]
            gbps = (pl['bytes'] / 1e9) / (pl['ms'] / 1e3) if pl['ms'] > 0 else 0.0
            return {'samples_per_s': B * n / dt, 'ms_per_step': 1e3 * dt / n, 'steps': n, 'first_step_s': first_s,
                    'plain_aggregation': {'launches': pl['launches'], 'avg_launch_us': 1e3 * pl['ms'] / max(1, pl['launches']), 'achieved': gbps,
                                          'unit': 'GB/s', 'frac': gbps / HBM_PEAK_GBPS,
                                          'entry_points': sorted(nm for nm in PLAIN_SPMM if got.get(nm, {}).get('tags', {}).get('plain'))}}
        if hip.operand_format == _l.FMT_F16X2:
            hip.operand_format = _l.FMT_BF16X3                        # (instance attribute over the class default)
            try:
                # the number to quote "at the reference's precision": the same model, same inputs, same --steps as the headline, every fp32 operand
                # of the matrix-core products as three bf16 pieces (24 significant bits = fp32's own)
                extras['alt_formats'] = {'bf16x3': side(model, As_in, n=a.steps)}
            finally:
                del hip.operand_format
        if not a.permute:
            t_p = time.perf_counter()
            graph_p = CsrGraph.queen_grid(a.grid, a.grid, normalize=True, permute_seed=1234, device=dev)
            model_p = M.STCGNN(N, C, a.order, a.order, 1, a.hidden, a.layers, a.pred, graph_mode=a.graph_mode, reorder_nodes=not a.no_reorder).to(dev)
            model_p.load_state_dict(model.state_dict())
            extras['permuted'] = side(model_p, graph_p)
            used = graph_p.with_locality()[0] if not a.no_reorder else graph_p
            extras['permuted'].update(what='SURVEY 8(d1) variant B: the queen grid under torch.Generator().manual_seed(1234) node order; the host renumbers it '
                                           '(reverse Cuthill-McKee) and plans patches as greedy clusters',
                                      setup_and_steps_s=time.perf_counter() - t_p, row_fetches_per_output_row=used.fetches_per_row[0],
                                      patch_stats={k: list(v) for k, v in getattr(used, 'patch_stats', {}).items()})
            del model_p, graph_p

    if rank == 0:
        total_ms = sum(d['ms'] for d in per_kernel.values()) or 1.0

        def gather(names, tag=None):
            acc = dict(launches=0, ms=0.0, bytes=0)
            for name in names:
                d = per_kernel.get(name)
                if d is None:
                    continue
                src = d if tag is None else d.get('tags', {}).get(tag)
                if src:
                    for key in acc:
                        acc[key] += src[key]
            return acc

        def rate(acc):
            return (acc['bytes'] / 1e9) / (acc['ms'] / 1e3) if acc['ms'] > 0 else 0.0

        plain = gather(PLAIN_SPMM, 'plain')                 # Y = S.X, no Y0: SURVEY 8(d3)'s byte formula verbatim
        every = gather(SPMM_ENTRY_POINTS)
        config_key = f'{a.storage}:{a.grid}:{C}:{a.hidden}:{B}:{a.order}:{a.layers}:{a.obs}:{a.pred}:{int(a.permute)}'
        traffic, traffic_note = pmc_traffic(a, config_key)
        achieved = rate(plain)
        roofline = {
            'bound': 'hbm',
            'kernel': ('spmm_patch_kernel (C-ABI stc_patch_spmm_' + a.storage + ': source rows of a cluster of 32 output rows staged through LDS)'
                       if 'stc_patch_spmm_' + a.storage in per_kernel else 'spmm_bcsr_kernel (C-ABI stc_bcsr_spmm_' + a.storage + ')')
                      + ': the plain aggregation launches Y = S.X of the timed train steps '
                      f'(S.state forward, S^T.dY backward; rows of C*hidden = {C * a.hidden} values, {B} samples per launch)',
            'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS,
            'traffic': traffic, 'traffic_source': 'committed' if traffic is not None else None, 'traffic_note': traffic_note,
            'launches': plain['launches'], 'avg_launch_us': 1e3 * plain['ms'] / max(1, plain['launches']),
            'algorithmic_bytes_per_launch': plain['bytes'] / max(1, plain['launches']),
            'bytes_formula': 'nnz*8 + 4*(N+1) + 2*B*N*F*sizeof(x)  (SURVEY 8(d3))',
            'aggregate': {'what': 'every aggregation launch of a step (two further steps with every launch timed; ' + ' + '.join(n for n in SPMM_ENTRY_POINTS if n in per_kernel)
                                  + '): plain, with the GRU blend in the epilogue, state-gradient sums, the two-ring forms (two aggregations per launch); graph once + every operand read once + every result written once',
                          'achieved': rate(every), 'frac': rate(every) / HBM_PEAK_GBPS, 'launches': every['launches'],
                          'avg_launch_us': 1e3 * every['ms'] / max(1, every['launches']),
                          'algorithmic_bytes_per_launch': every['bytes'] / max(1, every['launches']),
                          # the two-ring launches (two aggregations each): where most of the step's aggregation happens since round 5
                          'two_ring': {name: {'launches': per_kernel[name]['launches'],
                                              'avg_launch_us': 1e3 * per_kernel[name]['ms'] / max(1, per_kernel[name]['launches']),
                                              'algorithmic_bytes_per_launch': per_kernel[name]['bytes'] / max(1, per_kernel[name]['launches']),
                                              'achieved': rate(per_kernel[name]), 'frac': rate(per_kernel[name]) / HBM_PEAK_GBPS,
                                              'traffic': pmc_entry_traffic(name, config_key)}
                                       for name in ('stc_ring2_sum_f32', 'stc_ring2_blend_f32', 'stc_ring2_chain_f32') if name in per_kernel}},
        }
        if plain['launches'] == 0 and every['launches'] > 0:
            # order 3 on a graph with two-ring plans: every aggregation of the step is a chained two-ring launch (two aggregations each), no plain
            # Y = S.X is left to price -- the roofline entry is then the aggregation launches as a whole, on their own algorithmic bytes
            roofline.update(kernel='the aggregation launches of the step as a whole (' + ' + '.join(n for n in SPMM_ENTRY_POINTS if n in per_kernel)
                                   + '): no plain Y = S.X launch is left in this configuration; graph once + every operand read once + every result written once',
                            achieved=rate(every), frac=rate(every) / HBM_PEAK_GBPS, launches=every['launches'],
                            avg_launch_us=1e3 * every['ms'] / max(1, every['launches']),
                            algorithmic_bytes_per_launch=every['bytes'] / max(1, every['launches']),
                            bytes_formula='per launch: nnz*8 + 4*(N+1) once + every operand plane read once + every result plane written once')
        dom_name = max(per_kernel, key=lambda n: per_kernel[n]['ms']) if per_kernel else None
        if dom_name is not None:
            dk = per_kernel[dom_name]
            dom = {'entry_point': dom_name, 'share_of_kernel_time': dk['ms'] / total_ms,
                   'measured_in': 'the timed region' if (dom_name in PRICED_ENTRY_POINTS or graphed) else 'two further steps', 'launches': dk['launches'],
                   'avg_launch_us': 1e3 * dk['ms'] / max(1, dk['launches'])}
            if dk['bytes']:
                dom.update(algorithmic_bytes_per_launch=dk['bytes'] / max(1, dk['launches']), achieved=rate(dk), unit='GB/s', frac=rate(dk) / HBM_PEAK_GBPS,
                           traffic=pmc_entry_traffic(dom_name, config_key),
                           bytes_formula='every operand plane read once + every result plane written once (planes of batch*N*C*hidden*4 bytes; '
                                         'an accumulated plane is read and written)')
                for tag, tv in dk.get('tags', {}).items():
                    dom.setdefault('forms', {})[tag] = {'launches': tv['launches'], 'avg_launch_us': 1e3 * tv['ms'] / max(1, tv['launches']),
                                                        'achieved': rate(tv), 'frac': rate(tv) / HBM_PEAK_GBPS}
            if dom_name.startswith('stc_cell_small'):
                # the small-graph cell launches (one workgroup per sample) are bound by the exact-fp32 matrix pipe + the vector instructions
                # around it, not by HBM: price them in flops against the fp32 matrix peak, and against the share of the chip a launch can use
                h, Kk = a.hidden, a.order
                def flops(cin, bwd):
                    proj = 2.0 * N * C * (Kk * Kk * (cin + h)) * 3 * h
                    mix = 2.0 * N * C * C * (Kk - 1) * 3 * h
                    agg = 2.0 * graph.nnz * C * (cin + 2 * h)
                    return B * ((2 * proj + mix + 2.0 * graph.nnz * C * 2 * (cin + h)) if bwd else (proj + mix + agg))
                narrow, wide = a.obs, (a.layers - 1) * a.obs + a.layers * a.pred
                per_cell = (narrow * flops(1, dom_name.endswith('bwd_f32')) + wide * flops(h, dom_name.endswith('bwd_f32'))) / max(1, narrow + wide)
                # a cell step is ONE launch (a workgroup per sample) or, split over several workgroups per sample, one launch per phase (four)
                launches_per_cell = max(1, round(dk['launches'] / a.steps / max(1, narrow + wide)))
                cell_us = dom['avg_launch_us'] * launches_per_cell
                tf = per_cell / (cell_us * 1e-6) / 1e12
                cus = min(256, B * (hip.cell_small_splits(B, N * C) if launches_per_cell > 1 else 1))
                dom['matrix'] = {'what': 'algorithmic flops of a cell step (projections, category mix, aggregation; backward: dZ + dW) on '
                                         'v_mfma_f32_16x16x4_f32, averaged over the narrow (layer 0) and wide cells of the schedule',
                                 'flops_per_cell_step': per_cell, 'launches_per_cell_step': launches_per_cell, 'cell_step_us': cell_us,
                                 'achieved': tf, 'peak': FP32_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                 'frac': tf / FP32_MATRIX_PEAK_TFLOPS, 'compute_units_in_use': cus,
                                 'frac_of_the_units_in_use': tf / (FP32_MATRIX_PEAK_TFLOPS * cus / 256.0)}
                per_launch = per_cell / launches_per_cell
                if plain['launches'] == 0:                  # no aggregation launch in this step at all: the dominant kernel IS the roofline entry
                    roofline.update(bound='mfma', kernel=f'{dom_name}: one STC_Cell step in {launches_per_cell} launch(es), '
                                                         f'{cus} of 256 compute units at work, exact-fp32 matrix instructions',
                                    achieved=tf, peak=FP32_MATRIX_PEAK_TFLOPS, unit='TFLOP/s', frac=tf / FP32_MATRIX_PEAK_TFLOPS,
                                    launches=dk['launches'], avg_launch_us=dom['avg_launch_us'], algorithmic_flops_per_launch=per_launch,
                                    frac_of_the_units_in_use=dom['matrix']['frac_of_the_units_in_use'])
                    for key in ('algorithmic_bytes_per_launch', 'bytes_formula', 'aggregate'):
                        roofline.pop(key, None)
            roofline['dominant'] = dom
        roofline['mfma'] = pmc_mfma(config_key)
        proj = projection_rates(a, per_kernel, B * N, C)
        if proj is not None:
            roofline['mfma'] = dict(roofline['mfma'] or {}, projection=proj)
        if not a.no_unit_d3:
            roofline['unit_d3'] = spmm_unit_d3(graph, dev, C, 2 * a.hidden, torch.bfloat16 if a.storage == 'bf16' else torch.float32,
                                               reorder=not a.no_reorder)
        value = world * B * a.steps / elapsed
        out = {
            'metric': METRIC, 'value': value, 'unit': 'samples/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * elapsed / a.steps,
            'higher_is_better': True, 'scaling': 'strong' if a.global_batch else 'weak', 'vs_baseline': None, 'dtype': a.storage, 'data': 'synthetic',
            'n_ranks_seen': n_ranks_seen, 'hip_graph': bool(graphed),
            # what a first multi-GPU run needs to explain itself: every rank's own step time over the timed region (the value uses the
            # slowest), and how long each took to get there (process start -> resident model; warm-up incl. node renumbering and plans)
            'per_rank': {'ms_per_step_min': min(per_rank['ms_per_step']), 'ms_per_step_max': max(per_rank['ms_per_step']),
                         'ms_per_step': [round(v, 3) for v in per_rank['ms_per_step']],
                         'build_s_max': max(per_rank['build_s']), 'warmup_s_max': max(per_rank['warmup_s'])},
            'config': {'workload': f'full STC-GNN train step (fwd + ComboLoss + bwd + grad all-reduce + Adam), {a.graph_mode}'
                                   f'{" (MGP_Gen learned dense graphs, " + str(sum(p.numel() for p in model.parameters())) + " parameters)" if learned else ""}, '
                                   f'{a.grid}x{a.grid} queen grid N={N} nnz={graph.nnz}{" permuted" if a.permute else ""}, C={C}, '
                                   f'hidden={a.hidden}, K={a.order}, layers={a.layers}, T={a.obs}+{a.pred}' + (', bf16 state storage' if a.storage == 'bf16' else ''),
                       'global_batch': world * B, 'batch_per_gpu': B, 'parallelism': f'batch-shard x{world}',
                       **({'preset': a.preset} if a.preset else {}),
                       'operand_format': {0: 'bf16x3', 1: 'f16x2'}.get(getattr(hip, 'operand_format', None)),
                       'grad_bucket_bytes': bucket.nbytes, 'grad_large_bytes': bucket.large_nbytes, 'input_seeds': 'X, Y: Bernoulli(0.1635), torch seed 1000 + rank'},
            'roofline': roofline,
            'power': power,
            'step_breakdown': {'fwd_loss_bwd_ms': phases[0], 'grad_allreduce_ms': phases[1], 'adam_ms': phases[2],
                               'samples_per_s_excluding_optimizer': world * B / ((phases[0] + phases[1]) / 1e3) if phases[0] > 0 else None,
                               'ms_per_step_without_launch_events': untimed_ms,
                               'note': 'value / ms_per_step are the whole step (Adam included) with two HIP-event records around every PRICED launch (the plain aggregation and the dominant cell kernels); '
                                       'the phases are HIP events on the compute stream of rank 0; ms_per_step_without_launch_events re-times '
                                       f'{untimed_steps} steps with the launch timer off'},
            **extras,
            'kernels': {k: {'launches': d['launches'], 'ms_per_step': d['ms'] / a.steps, 'share': d['ms'] / total_ms,
                            **({'GBps': d['bytes'] / 1e9 / (d['ms'] / 1e3)} if d['bytes'] and d['ms'] > 0 else {})}
                        for k, d in sorted(per_kernel.items(), key=lambda kv: -kv[1]['ms'])},
            'loss': float(loss.detach()),
            'hbm_peak_allocated_gb': torch.cuda.max_memory_allocated(dev) / 1e9,
            'hbm_peak_reserved_gb': torch.cuda.max_memory_reserved(dev) / 1e9,
        }
        fmt_name = out['config']['operand_format']
        if 'alt_formats' in extras:
            g24 = extras['alt_formats']['bf16x3']
            out['value_fp32_grade'] = {'value': g24['samples_per_s'], 'unit': 'samples/s', 'ms_per_step': g24['ms_per_step'], 'steps': g24['steps'],
                                       'operand_format': 'bf16x3',
                                       'what': 'the same train step with every fp32 operand of the matrix-core products held to 24 significant bits (three bf16 pieces, six '
                                               "products): the figure to compare at the reference's own fp32 precision; `value` runs the default format " + str(fmt_name)}
        elif fmt_name == 'bf16x3' and a.storage == 'f32':
            out['value_fp32_grade'] = {'value': value, 'unit': 'samples/s', 'ms_per_step': out['ms_per_step'], 'steps': a.steps, 'operand_format': 'bf16x3',
                                       'what': '`value` itself: this run uses the 24-bit operand format'}
        out['dtype_detail'] = ('bf16 state storage, fp32 parameters and sums' if a.storage == 'bf16' else
                               'fp32 planes, parameters and sums; aggregations in fp32 fmaf; matrix-core products on fp32 operands split as '
                               + {'f16x2': 'two fp16 pieces (22 significant bits)', 'bf16x3': 'three bf16 pieces (24 significant bits)'}.get(fmt_name, str(fmt_name)))
        if world == 1 and not a.no_cpu_baseline:
            if N <= 1024:                                        # small graphs: the whole model through the oracle's dense (reference) algorithm
                out['cpu_baseline'] = cpu_baseline_small(a, graph.to_dense(), Gc_cpu, sd_cpu,
                                                         As_dense=CsrGraph.queen_grid(a.grid, a.grid, normalize=False).to_dense() if learned else None)
            else:
                out['cpu_baseline'] = cpu_baseline(a, _sparse_T(graph), Gc_cpu, sd_cpu)
        if world == 1 and a.preset is None and not (a.no_presets or a.no_extras or graphed or learned or a.permute) and a.storage == 'f32':
            # BASELINE.json's other configurations in the same line: the metric's memory goes back first (a preset is a process of its own)
            del loss
            model.zero_grad(set_to_none=True)
            model, opt, bucket, X, Y, As_in, graph = None, None, None, None, None, None, None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out['presets'] = run_presets(torch.cuda.memory_reserved(dev) / 1e9)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def bench_cfg2(a):
    """BASELINE configuration 2: ONE BDG_Dif layer (reference STC_GNN.py:31-47) at B = 32, N = 200 (10 x 20 queen grid), C = 8, L = 32,
    Ho = 32, Chebyshev order ``--order``, dense learned-style Gs (full N x N pattern, differentiable) -- forward and forward + backward
    through the drop-in module, HIP events over ``--steps`` repetitions after ``--warmup``; BASELINE.md section 2 holds the reference's
    CPU times for this shape (13.3 / 25.8 ms at K = 2, 29.6 / 55.2 ms at K = 3 on 8 cores)."""
    for _p in (REPO, os.path.join(REPO, 'stc-gnn_amd')):
        if _p not in sys.path:
            sys.path.insert(0, _p)
    import torch
    import STC_GNN as M
    dev = torch.device('cuda', 0)
    B, N, C, L, Ho, K = 32, 200, 8, 32, 32, a.order
    g = torch.Generator().manual_seed(0)
    torch.manual_seed(42)
    layer = M.BDG_Dif(K, K, L, Ho).to(dev)
    Gs = torch.softmax(torch.randn(N, N, generator=g), -1).to(dev).requires_grad_()
    Gc = torch.softmax(torch.randn(C, C, generator=g), -1).to(dev).requires_grad_()
    X = torch.randn(B, N, C, L, generator=g).to(dev).requires_grad_()
    R = torch.randn(B, N, C, Ho, generator=g).to(dev)
    steps, warm = max(a.steps, 20), max(a.warmup, 3)

    def timed(fn):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        rounds = []                                              # three rounds of ``steps``, the median one: a round is a few ms, and one
        for _ in range(3):                                       # host hiccup (allocator, clock ramp of a fresh box) in it measured 5x
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(steps):
                fn()
            e.record()
            torch.cuda.synchronize()
            rounds.append(s.elapsed_time(e) / steps)
        return sorted(rounds)[1]

    def fwd():
        with torch.no_grad():
            return layer(X, Gs, Gc)

    def fwd_bwd():
        for t in (X, Gs, Gc, *layer.parameters()):
            t.grad = None
        (layer(X, Gs, Gc) * R).sum().backward()

    ms_f, ms_fb = timed(fwd), timed(fwd_bwd)
    eager = dict(forward_ms=ms_f, forward_backward_ms=ms_fb)
    graphed = bool(a.hip_graph)
    if graphed:
        # the same two callables captured once each into a HIP graph (after the eager runs above: every plan, workspace and table exists) and
        # replayed: the gradients land in the capture's own buffers, re-used by every replay
        def capture(fn):
            torch.cuda.synchronize()
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                fn()
            return cg.replay
        try:
            ms_f, ms_fb = timed(capture(fwd)), timed(capture(fwd_bwd))
        except RuntimeError as e:                                  # (something on the path synchronised with the host: the eager numbers stand)
            graphed, eager['capture_error'] = False, str(e).split('\n')[0][:200]
    ref = {2: (13.3, 25.8), 3: (29.6, 55.2)}.get(K)
    cpu = None
    if not a.no_cpu_baseline:                                    # the oracle's bdg_dif (the reference's algorithm op for op) on the host cores
        from oracle import stc_oracle as O
        threads = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(threads)
        warm_c, timed_c = (int(v) for v in a.cpu_shots.split(','))
        W_, b_ = (p.detach().cpu().clone().requires_grad_() for p in (layer.W, layer.b))
        Xc, Gsc, Gcc, Rc = X.detach().cpu().requires_grad_(), Gs.detach().cpu().requires_grad_(), Gc.detach().cpu().requires_grad_(), R.cpu()
        tf, tfb = [], []
        for shot in range(warm_c + timed_c):
            t0 = time.perf_counter()
            with torch.no_grad():
                O.bdg_dif(Xc, Gsc, Gcc, W_, b_, K, K)
            tf.append(time.perf_counter() - t0)
            for t_ in (Xc, Gsc, Gcc, W_, b_):
                t_.grad = None
            t0 = time.perf_counter()
            (O.bdg_dif(Xc, Gsc, Gcc, W_, b_, K, K) * Rc).sum().backward()
            tfb.append(time.perf_counter() - t0)
        cpu = dict(value=1.0 / _median(tfb[warm_c:]), unit='layers/s', cores=threads, kind='port',
                   sample=f'oracle.bdg_dif (reference algorithm op for op) at the same shape, {warm_c} warm-up + {timed_c} timed, median: forward '
                          f'{1e3 * _median(tf[warm_c:]):.2f} ms, forward + backward {1e3 * _median(tfb[warm_c:]):.2f} ms',
                   forward_ms=1e3 * _median(tf[warm_c:]), forward_backward_ms=1e3 * _median(tfb[warm_c:]))
    print(json.dumps({
        'metric': 'BDG_Dif layer forward+backward layer-applications/sec (BASELINE configuration 2; not the headline metric)', 'value': 1e3 / ms_fb,
        'unit': 'layers/s', 'n_gpus': 1, 'steps': steps, 'warmup': warm, 'ms_per_step': ms_fb, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'single BDG_Dif layer, B={B} N={N} C={C} L={L} Ho={Ho} K={K}, dense differentiable Gs and Gc (gradients of X, W, b, Gs, Gc)',
                   'preset': 'cfg2'},
        'forward_ms': ms_f, 'forward_backward_ms': ms_fb, 'hip_graph': graphed, 'eager': eager,
        'timing': f'HIP events over three rounds of {steps} repetitions each, the median round' + ('; replays of the captured HIP graph (`eager`: per-launch dispatch)' if graphed else ''),
        'cpu_baseline': cpu,
        'roofline': {'bound': 'mfma', 'kernel': 'stc_dense_agg_f32 + the node kernels of one BDG_Dif (exact-fp32 matrix instructions)',
                     'achieved': (2.0 * B * N * N * C * L * (K - 1) + 2.0 * B * N * C * (K * K * L) * Ho + 2.0 * B * N * C * C * (K - 1) * Ho) / (ms_f * 1e-3) / 1e12,
                     'peak': FP32_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': (2.0 * B * N * N * C * L * (K - 1) + 2.0 * B * N * C * (K * K * L) * Ho + 2.0 * B * N * C * C * (K - 1) * Ho) / (ms_f * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
                     'traffic': None, 'what': 'algorithmic flops of the FORWARD (dense aggregation on the features, projection, category mix) / its time'},
        'reference_cpu_ms': None if ref is None else {'forward': ref[0], 'forward_backward': ref[1], 'cores': 8,
                                                      'source': 'BASELINE.md section 2 (reference imported unmodified in the survey container)'},
    }), flush=True)


def _sparse_T(graph):
    """torch sparse CSR of Gs^T straight from the graph's forward operand (no dense N x N detour)."""
    import torch
    h = graph._host
    return torch.sparse_csr_tensor(torch.from_numpy(h['fwd_rowptr']).long(), torch.from_numpy(h['fwd_colidx']).long(),
                                   torch.from_numpy(h['fwd_val']), size=(graph.n, graph.n))


if __name__ == '__main__':
    main()
