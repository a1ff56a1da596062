"""Training / evaluation harness counterpart of the reference's ``framework/Model_Trainer.py:26-158``.

Same protocol: ``STCGNN`` built from the same ``params`` keys, ComboLoss, Adam(lr, weight_decay), epoch loop over
the modes, best-validation checkpoint ``{epoch, train_loss, val_loss, state_dict}`` written to
``<output_dir>/<model>-<time_slice>.pkl``, early stopping, ``test()`` from that checkpoint.  Deliberate
differences: the running loss is accumulated as a Python float (the reference adds the loss TENSOR and so keeps
every step's autograd graph alive: +0.7 GB per step, ``Model_Trainer.py:85``); evaluation runs under
``no_grad`` (``:136-138`` does not); no ``empty_cache()`` per step.  Checkpoints interchange with the reference.

Batch-sharded training: when a process group is initialised (``stc_hip.dist.init_from_env`` under
``torchrun``, one rank per GPU) every rank holds the whole window set, takes its contiguous shard of each batch
(``shard_batch``), and the parameter gradients -- views into ONE flat ``GradBucket`` -- are summed by a single RCCL
all-reduce per step, each rank weighted by its share of the batch (so a ragged last batch stays exact).  Losses are
reduced the same way; rank 0 writes the checkpoint.  With learned graphs the model must be built ``batch_sharded=True``
(done here): MGP_Gen sums over the batch before its softmax.  One rank: the same code, no collective.
"""
from __future__ import annotations

import os
import time
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

from . import dist as sdist
from .loss import ComboLoss
from .optim import Adam as StcAdam


class Trainer:
    """``hip_graph=True`` (one rank): after two eager steps per batch shape the whole train step -- forward, ComboLoss, backward, Adam --
    is captured into a HIP graph (``torch.cuda.graph``; every kernel of this path launches on the capturing stream and allocates only
    through torch) and replayed for every later batch of that shape from static input buffers.  At the SF shape the step is ~700 launches
    of a few microseconds each, i.e. launch-bound: replay halves it (HISTORY.md, section 5); same parameters as eager, bit for bit."""

    def __init__(self, params: dict, data: dict, graph_mode: str = 'dense-learned', hip_graph: bool = False):
        from STC_GNN import STCGNN
        self.params = params
        dev = params['device']
        if not str(dev).startswith('cuda') and not params.get('_allow_cpu_for_tests'):
            raise ValueError("device must be 'cuda:k': the STC-GNN hot path has no CPU implementation")
        self.mask = data.get('mask')
        self.threshold = data.get('HA')
        from .graph import CsrGraph
        s_adj = data['s_adj']                                    # the reference's dense (N, N) array, or a CsrGraph (csr-fixed, any N)
        self.prior_graph = [s_adj if isinstance(s_adj, CsrGraph) else torch.from_numpy(np.asarray(s_adj)).float().to(dev),
                            torch.from_numpy(np.asarray(data['c_cor'])).float().to(dev)]
        if isinstance(s_adj, CsrGraph) and graph_mode != 'csr-fixed':
            raise ValueError("a CsrGraph prior graph needs graph_mode='csr-fixed' (the learned-graph mode mixes a dense prior)")
        if params.get('model', 'STC-GNN') != 'STC-GNN':
            raise NotImplementedError('Invalid model name.')
        import torch.distributed as tdist
        self.rank = tdist.get_rank() if tdist.is_available() and tdist.is_initialized() else 0
        self.world = tdist.get_world_size() if tdist.is_available() and tdist.is_initialized() else 1
        self.model = STCGNN(num_nodes=params['H'] * params['W'], num_categories=params['C'],
                            Ks=params['cheby_order'], Kc=params['cheby_order'], input_dim=1,
                            hidden_dim=params['hidden_dim'], num_layers=params['nn_layers'],
                            out_horizon=params['pred_len'], graph_mode=graph_mode,
                            batch_sharded=self.world > 1 and graph_mode == 'dense-learned').to(dev)
        sdist.broadcast_parameters(self.model)                          # every replica starts from rank 0's draw
        self.criterion = ComboLoss()
        self.bucket = sdist.GradBucket(self.model.parameters())        # every .grad is a view into one flat buffer
        self.hip_graph = bool(hip_graph) and self.world == 1 and str(dev).startswith('cuda')
        # torch.optim.Adam, with parameters of >= 64 MiB (MixedFusion's two matrices in the learned-graph mode) updated by stc_adam_f32 (optim.py)
        self.optimizer = StcAdam(self.model.parameters(), lr=params['learn_rate'], weight_decay=params['decay_rate'],
                                 capturable=self.hip_graph)
        self._captured = {}                                           # batch shape -> [eager steps seen, graph, static x, static y, static loss]
        self._capture_shape = None                                    # only the FIRST batch shape seen (the full batch) is captured
        self._capture_stream = torch.cuda.Stream(dev) if self.hip_graph else None

    @property
    def checkpoint_path(self) -> str:
        return os.path.join(self.params['output_dir'], f'{self.params.get("model", "STC-GNN")}-{self.params["time_slice"]}.pkl')

    def _forward(self, x_seq):
        return self.model(X_seq=x_seq, As=self.prior_graph[0], Ac=self.prior_graph[1])

    def _step(self, x_seq, y_true):
        """One whole-batch train step on this rank (no collective): the body that ``hip_graph`` captures."""
        self.bucket.zero()
        loss = self.criterion(self._forward(x_seq), y_true)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def _graphed_step(self, x_seq, y_true):
        key = (tuple(x_seq.shape), tuple(y_true.shape))
        if self._capture_shape is None:
            self._capture_shape = key
        if key != self._capture_shape:                                # e.g. the ragged last batch of an epoch: a graph of its own would
            return self._step(x_seq, y_true)                          # hold a second private pool of activations; it runs eagerly
        slot = self._captured.setdefault(key, [0, None, None, None, None])
        if slot[1] is None:
            if slot[0] < 2:                                           # eager warm-up steps (they are real steps), run on the stream the
                slot[0] += 1                                          # capture will use: optimizer state, cached graph operands and that
                cs = self._capture_stream                             # stream's kernel workspace all exist before the capture begins
                cs.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cs):
                    loss = self._step(x_seq, y_true)
                torch.cuda.current_stream().wait_stream(cs)
                return loss
            slot[2], slot[3] = x_seq.clone(), y_true.clone()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=self._capture_stream):
                slot[4] = self._step(slot[2], slot[3])
            slot[1] = graph
        slot[2].copy_(x_seq)
        slot[3].copy_(y_true)
        slot[1].replay()
        return slot[4]

    def train(self, data_loader: Dict[str, Iterable], modes: List[str] = ('train', 'validate'),
              early_stop_patience: int = 10, verbose: bool = True) -> dict:
        history = {m: [] for m in modes}
        run_time = {m: [] for m in modes}
        best_val, patience = np.inf, early_stop_patience
        os.makedirs(self.params['output_dir'], exist_ok=True)
        for epoch in range(1, 1 + self.params['num_epochs']):
            for mode in modes:
                self.model.train(mode == 'train')
                total, seen, t0 = 0.0, 0, time.time()
                for x_seq, y_true in data_loader[mode]:
                    n = y_true.shape[0]
                    # fewer samples than ranks (a short last batch): decided from n, which every rank knows, BEFORE any collective --
                    # every rank then runs the whole small batch (same gradient on every rank, so their mean is that gradient)
                    replicated = self.world > 1 and n < self.world
                    if self.world > 1 and not replicated:                # this rank's contiguous shard of the batch
                        x_seq = sdist.shard_batch(x_seq, self.rank, self.world, ragged=True)
                        y_true = sdist.shard_batch(y_true, self.rank, self.world, ragged=True)
                    share = 1.0 / self.world if replicated else y_true.shape[0] / n
                    mgp = getattr(self.model, 'mix_graph_pair', None)
                    sharded_graphs = bool(mgp is not None and getattr(mgp, 'batch_sharded', False))
                    if replicated and sharded_graphs:
                        mgp.batch_sharded = False                        # the batch sum is already complete on every rank
                    if mode == 'train' and self.hip_graph:
                        total += float(self._graphed_step(x_seq, y_true)) * n
                        seen += n
                        continue
                    with torch.set_grad_enabled(mode == 'train'):
                        if mode == 'train':
                            self.bucket.zero()                           # zero_grad() that keeps the bucket views
                        try:
                            loss = self.criterion(self._forward(x_seq), y_true)
                            if mode == 'train':
                                # unequal shards (ragged last batch): weight by this rank's share so that the mean over
                                # ranks of the bucket is the full-batch gradient; equal shards: the factor is exactly 1
                                (loss if share * self.world == 1 else loss * (share * self.world)).backward()
                            loss = loss.detach()
                        finally:
                            if replicated and sharded_graphs:
                                mgp.batch_sharded = True
                        if mode == 'train':
                            self.bucket.allreduce_mean()                 # the one collective of a step (no-op on one rank)
                            self.optimizer.step()
                    total += float(self._global_loss(loss, share)) * n   # a float: no graph is kept alive
                    seen += n
                run_time[mode].append(time.time() - t0)
                history[mode].append(total / max(seen, 1))
            val = history['validate'][-1] if 'validate' in history else history['train'][-1]
            if val < best_val:
                best_val, patience = val, early_stop_patience
                if self.rank == 0:
                    torch.save({'epoch': epoch, 'train_loss': history['train'][-1], 'val_loss': val,
                                'state_dict': self.model.state_dict()}, self.checkpoint_path)
                note = 'checkpoint updated'
            else:
                patience -= 1
                note = f'no improvement ({patience} left)'
            if verbose and self.rank == 0:
                print(f'Epoch {epoch}: train {history["train"][-1]:.4f} ({run_time["train"][-1]:.2f} s), '
                      f'validate {val:.4f}; {note}')
            if patience == 0:
                if verbose and self.rank == 0:
                    print(f'Early stopping triggered at epoch {epoch}.')
                break
        if self.world > 1:
            import torch.distributed as tdist
            tdist.barrier()                                              # the checkpoint is on disk before any rank tests
        return dict(loss=history, seconds=run_time, best_val=best_val)

    def _global_loss(self, loss: torch.Tensor, share: float) -> torch.Tensor:
        """The full-batch loss from the shard losses: ComboLoss is a mean over samples / elements, so it is the
        share-weighted sum over ranks (every rank gets the same value, hence the same early-stopping decisions)."""
        if self.world == 1:
            return loss
        import torch.distributed as tdist
        t = (loss * share).reshape(1).clone()
        tdist.all_reduce(t)
        return t[0]

    @torch.no_grad()
    def test(self, data_loader: Dict[str, Iterable], modes: List[str] = ('test',), checkpoint: Optional[str] = None) -> dict:
        ckpt = torch.load(checkpoint or self.checkpoint_path, map_location=self.params['device'])
        self.model.load_state_dict(ckpt['state_dict'])
        self.model.eval()
        out = {}
        from . import metrics as smetrics
        for mode in modes:
            pred, truth = [], []
            # every rank evaluates the WHOLE batches (the forecasts are wanted whole on rank 0, and evaluation is a forward pass only):
            # the learned graphs' batch sum is then complete on each rank and must not be all-reduced again
            mgp = getattr(self.model, 'mix_graph_pair', None)
            sharded_graphs = bool(mgp is not None and getattr(mgp, 'batch_sharded', False))
            if sharded_graphs:
                mgp.batch_sharded = False
            try:
                with torch.no_grad():
                    for x_seq, y_true in data_loader[mode]:
                        pred.append(self._forward(x_seq))
                        truth.append(y_true)
            finally:
                if sharded_graphs:
                    mgp.batch_sharded = True
            pred_d, truth_d = torch.cat(pred, 0), torch.cat(truth, 0)        # stay on the device for the evaluation
            pred, truth = pred_d.cpu().numpy(), truth_d.cpu().numpy()
            eps = 1e-7
            bce = float(-(truth * np.log(np.clip(pred, eps, 1)) + (1 - truth) * np.log(np.clip(1 - pred, eps, 1))).mean())
            out[mode] = dict(forecast=pred, ground_truth=truth, bce=bce, mae=float(np.abs(pred - truth).mean()),
                             epoch=ckpt['epoch'])
            if self.threshold is not None:
                # the reference's evaluation protocol (Model_Trainer.py:148-154): masked cells dropped, per-step metrics
                H, W = self.params['H'], self.params['W']
                mask = None if self.mask is None or len(self.mask) == 0 else [tuple(int(v) for v in m) for m in np.asarray(self.mask).reshape(-1, 2)]
                out[mode]['metrics'] = smetrics.evaluate_binary(smetrics.mask_data(pred_d, H, W, mask),
                                                                 smetrics.mask_data(truth_d, H, W, mask), list(np.asarray(self.threshold)))
                if self.rank == 0:                                   # the reference's CSV log, same file name and layout (Metrics.py:54-85)
                    log_params = {k: v for k, v in self.params.items() if not k.startswith('_')}
                    smetrics.append_metrics_csv(os.path.join(self.params['output_dir'], f'{self.params.get("model", "STC-GNN")}_eval-bi-metrics.csv'),
                                                log_params, mode, out[mode]['metrics'])
        return out
