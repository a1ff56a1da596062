"""``torch.optim.Adam`` with the LARGE parameters on a streaming HIP kernel.

The reference's harness step (Model_Trainer.py:71-87) is ``Adam(lr, weight_decay)`` over a model whose two ``MixedFusion`` matrices hold
99.99 % of the parameters (2 x 400 MB at the SF shape).  torch's fused multi-tensor Adam moves their 5.6 GB per step at 4.4 TB/s;
``stc_adam_f32`` is one launch per tensor (include/stc_hip.h).  Everything else -- small parameters, CPU tensors, amsgrad / maximize /
differentiable groups, other dtypes -- stays with torch's own implementation: this class only takes the large tensors out of its way.
State entries have torch's names (``step``, ``exp_avg``, ``exp_avg_sq``), so ``state_dict`` round-trips with ``torch.optim.Adam``;
``step`` of a large parameter is a device scalar (as with ``capturable=True``): the update can be captured into a HIP graph.
"""
from __future__ import annotations

import torch

from .ops import kernels


class Adam(torch.optim.Adam):
    LARGE_BYTES = 64 << 20       # as dist.GradBucket: parameters from this size on are handled one by one

    def _ours(self, group, p) -> bool:
        return (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.numel() * 4 >= self.LARGE_BYTES and p.numel() % 4 == 0
                and p.data_ptr() % 16 == 0 and not (group.get('amsgrad') or group.get('maximize') or group.get('differentiable'))
                and p.grad is not None and not p.grad.is_sparse and p.grad.is_contiguous() and p.grad.data_ptr() % 16 == 0)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        large = [(group, p, p.grad) for group in self.param_groups for p in group['params'] if self._ours(group, p)]
        for _, p, _ in large:
            p.grad = None                                            # torch's step skips parameters without a gradient
        try:
            # the undecorated implementation: torch wraps every optimizer's step in its hook / profiler wrapper, and calling the wrapped parent from this
            # (also wrapped) step would fire the registered step hooks twice
            getattr(torch.optim.Adam.step, '__wrapped__', torch.optim.Adam.step)(self)
        finally:
            for _, p, g in large:
                p.grad = g
        for group, p, g in large:
            st = self.state[p]
            if len(st) == 0:
                st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if not st['step'].is_cuda or st['step'].dtype != torch.float32:   # (a state dict loaded from a non-capturable torch.optim.Adam)
                st['step'] = st['step'].to(device=p.device, dtype=torch.float32)
            st['step'] += 1
            beta1, beta2 = group['betas']
            kernels().adam(p, g, st['exp_avg'], st['exp_avg_sq'], st['step'], group['lr'], beta1, beta2, group['eps'], group['weight_decay'])
        return loss
