import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'stc-gnn_amd'))
import torch, numpy as np
from stc_hip import data as sdata
from stc_hip.trainer import Trainer
params = dict(device='cuda:0', H=10, W=10, C=5, batch_size=32, obs_len=9, pred_len=3, split_ratio=[6,1,1], model='STC-GNN', cheby_order=2, hidden_dim=16, nn_layers=2,
              learn_rate=2e-3, decay_rate=1e-4, num_epochs=2, time_slice=4, city='SYN')
data = sdata.synthetic_incidents(10, 10, 5, 500, sparse_graph=True)
def run(graphed, tag):
    loaders = sdata.get_data_loader(params, data, 9, 3, [6,1,1])
    torch.manual_seed(1)
    t = Trainer(dict(params, output_dir='/tmp/gd_' + tag), data, graph_mode='csr-fixed', hip_graph=graphed)
    h = t.train(loaders, verbose=False)
    return torch.cat([p.detach().flatten() for p in t.model.parameters()]).cpu(), h['loss']['train']
a, la = run(False, 'a'); b, lb = run(False, 'b'); c, lc = run(True, 'c'); d, ld = run(True, 'd')
print('eager vs eager', float((a-b).abs().max()), la, lb)
print('eager vs graph', float((a-c).abs().max()), lc)
print('graph vs graph', float((c-d).abs().max()), ld)
