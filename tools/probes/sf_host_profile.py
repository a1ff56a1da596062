import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'stc-gnn_amd'))
import torch, STC_GNN as M
from stc_hip import CsrGraph, dist as sdist
from stc_hip.loss import ComboLoss
dev = torch.device('cuda', 0)
N, C, B = 100, 5, 32
graph = CsrGraph.queen_grid(10, 10, normalize=True, device=dev)
torch.manual_seed(42)
model = M.STCGNN(N, C, 2, 2, 1, 16, 2, 3, graph_mode='csr-fixed').to(dev)
Gc = torch.softmax(torch.randn(C, C), -1).to(dev)
X = (torch.rand(B, 9, N, C) < 0.16).float().to(dev); Y = (torch.rand(B, 3, N, C) < 0.16).float().to(dev)
crit = ComboLoss(); bucket = sdist.GradBucket(model.parameters()); opt = torch.optim.Adam(model.parameters(), lr=2e-3)
def step():
    bucket.zero(); loss = crit(model(X_seq=X, As=graph, Ac=Gc), Y); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print('ms/step', (time.perf_counter() - t) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
