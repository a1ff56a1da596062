"""Probe (MI355X): the encoder-decoder's prediction and every parameter gradient under the two operand formats of the split-operand
matrix-core kernels (STC_OPERAND_FORMAT), against the CPU oracle in float64, on a small grid with the real loss (tiny gradients)."""
import os, sys, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'stc-gnn_amd'))
    import torch
    import STC_GNN as M
    from stc_hip import CsrGraph
    from oracle import stc_oracle as O
    G, C, h, K, B, T, hor = int(os.environ.get('G', 24)), 32, 16, 2, 2, 3, 2
    N = G * G
    graph = CsrGraph.queen_grid(G, G, normalize=True)
    torch.manual_seed(42)
    model = M.STCGNN(N, C, K, K, 1, h, 2, hor, graph_mode='csr-fixed')
    g = torch.Generator().manual_seed(8)
    Gc = torch.softmax(torch.randn(C, C, generator=g), -1)
    X = (torch.rand(B, T, N, C, generator=g) < 0.1635).float()
    Y = (torch.rand(B, hor, N, C, generator=g) < 0.1635).float()
    sd = {k: v.double().clone().requires_grad_() for k, v in model.state_dict().items()}
    hh = graph._host
    ST = torch.sparse_csr_tensor(torch.from_numpy(hh['fwd_rowptr']).long(), torch.from_numpy(hh['fwd_colidx']).long(), torch.from_numpy(hh['fwd_val']),
                                 size=(N, N)).double()
    want = O.encdec_forward(X.double(), ST, Gc.double(), sd, K, K, h, 2, hor, conv=O.bdg_dif_sparse)
    loss_w = O.combo_loss(want, Y.double()); loss_w.backward()
    dev = torch.device('cuda')
    model = model.to(dev)
    got = model(X_seq=X.to(dev), As=graph, Ac=Gc.to(dev))
    loss = O.combo_loss(got, Y.to(dev)); loss.backward(); torch.cuda.synchronize()
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-300))
    out = {'fwd': rel(got, want.detach())}
    for n, p in model.named_parameters():
        out[n] = rel(p.grad, sd[n].grad) if p.grad is not None else None
    print(json.dumps(out))
else:
    for fmt in ('bf16x3', 'f16x2'):
        r = subprocess.run([sys.executable, __file__, 'child'], env={**os.environ, 'STC_OPERAND_FORMAT': fmt}, capture_output=True, text=True)
        line = [l for l in r.stdout.split('\n') if l.startswith('{')]
        if not line:
            print(fmt, 'FAILED', r.stderr[-2000:]); continue
        d = json.loads(line[0])
        print(fmt, ' fwd %.2e' % d.pop('fwd'), ' worst grad %.2e' % max(v for v in d.values() if v is not None))
        for k, v in d.items():
            print('    %-40s %.2e' % (k, v))
