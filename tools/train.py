#!/usr/bin/env python3
"""CLI counterpart of the reference's ``framework/Main.py``: same flags, MI355X path.

    python tools/train.py -city SF -in /path/to/data -epoch 5            # learned graphs, reference semantics
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py -city SF ...
                                                                         # batch-sharded: one rank per GPU, RCCL all-reduce
"""
import argparse
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'stc-gnn_amd')):
    sys.path.insert(0, p)
from stc_hip import data as sdata          # noqa: E402
from stc_hip import dist as sdist          # noqa: E402
from stc_hip.trainer import Trainer        # noqa: E402

CITY = {'SF': dict(C=5, H=10, W=10, time_slice=4), 'NYC': dict(C=8, H=20, W=15, time_slice=6), 'CHI': dict(C=4, H=10, W=24, time_slice=4)}

ap = argparse.ArgumentParser(description='Run multi-incident co-prediction on MI355X.')
ap.add_argument('-device', '--device', default='cuda:0')
ap.add_argument('-in', '--input_dir', default='../data')
ap.add_argument('-out', '--output_dir', default='./output')
ap.add_argument('-city', '--city', default='SF', choices=sorted(CITY))
ap.add_argument('-obs', '--obs_len', type=int, default=9)
ap.add_argument('-pred', '--pred_len', type=int, default=3)
ap.add_argument('-split', '--split_ratio', type=int, nargs='+', default=[6, 1, 1])
ap.add_argument('-batch', '--batch_size', type=int, default=32)
ap.add_argument('-hidden', '--hidden_dim', type=int, default=16)
ap.add_argument('-K', '--cheby_order', type=int, default=2)
ap.add_argument('-nn', '--nn_layers', type=int, default=2)
ap.add_argument('-lr', '--learn_rate', type=float, default=2e-3)
ap.add_argument('-dr', '--decay_rate', type=float, default=1e-4)
ap.add_argument('-epoch', '--num_epochs', type=int, default=100)
ap.add_argument('-test', '--test_only', type=int, default=0, choices=[0, 1])
# beyond Main.py: fixed sparse graphs (any N) and a synthetic incident series, so that the trainer runs at sizes the SF file does not have
ap.add_argument('-graph', '--graph_mode', default='dense-learned', choices=['dense-learned', 'csr-fixed'],
                help="dense-learned = the reference's learned graphs (MGP_Gen, N <~ 300); csr-fixed = the prior graph used as given, row-normalised")
ap.add_argument('-synthetic', '--synthetic', type=int, nargs=4, metavar=('H', 'W', 'C', 'T'), default=None,
                help='train on a Bernoulli(0.1635) incident series of T steps on an H x W queen grid with C categories instead of a city file')
ap.add_argument('-stride', '--time_slice', type=int, default=None, help=argparse.SUPPRESS)
ap.add_argument('-hipgraph', '--hip_graph', type=int, default=0, choices=[0, 1], help='replay the train step as a captured HIP graph (one rank; launch-bound shapes)')
params = vars(ap.parse_args())
rank, world, local = sdist.init_from_env()                      # under torchrun: one rank per GPU, each on its own device
if world > 1:
    params['device'] = f'cuda:{local}'
syn = params.pop('synthetic')
ts = params.pop('time_slice')
params.update(CITY[params['city']], model='STC-GNN')
params['output_dir'] = os.path.join(params['output_dir'], params['city'] if syn is None else 'synthetic')
if syn is None:
    data = sdata.load_incidents(os.path.join(params['input_dir'], f'{params["city"]}-incidents-{params["time_slice"]}h.npz'))
else:
    params.update(H=syn[0], W=syn[1], C=syn[2], time_slice=ts or 4)
    data = sdata.synthetic_incidents(*syn, sparse_graph=params['graph_mode'] == 'csr-fixed')
loaders = sdata.get_data_loader(params, data, params['obs_len'], params['pred_len'], params['split_ratio'])
trainer = Trainer(params, data, graph_mode=params.pop('graph_mode'), hip_graph=bool(params.pop('hip_graph')))
if not params['test_only']:
    trainer.train(loaders)
res = trainer.test(loaders)
if rank == 0:
    print({m: {k: v for k, v in r.items() if k in ('bce', 'mae', 'epoch')} for m, r in res.items()})
