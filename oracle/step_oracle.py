"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the optimizer step of the reference's training loop.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module; the product path
(prcv2025reid_amd/trainer.py + csrc/optim.hip) never does.

Restates, in plain fp32 torch/numpy on the CPU:
  * _sanitize_grads                      train.py:85-96
  * the adaptive / fixed clip rule       train.py:981-1008 (np.percentile(last 10, 70) * 1.15 clamped to [0.5, 3])
  * torch.nn.utils.clip_grad_norm_       as called at train.py:1001 / 1006-1008
  * torch.optim.AdamW (single-tensor form, no amsgrad) as constructed at train.py:1460
  * the warm-up + cosine LambdaLR lambda train.py:1249-1262
Pinned by tests/test_step_cpu.py against torch.optim.AdamW / clip_grad_norm_ / LambdaLR themselves (the library calls the
reference makes) and against hand-computed percentile cases.
"""
import math
from typing import Dict, List

import numpy as np
import torch


def sanitize_grads(grads: List[torch.Tensor]) -> int:
    bad_total = 0
    for g in grads:
        bad = ~torch.isfinite(g)
        bad_total += int(bad.sum())
        g[bad] = 0.0
    return bad_total


def total_norm(grads: List[torch.Tensor]) -> float:
    return math.sqrt(sum(float(g.double().pow(2).sum()) for g in grads))


def adaptive_max_norm(history: List[float]) -> float:
    if len(history) > 10:
        return float(min(3.0, max(0.5, np.percentile(history[-10:], 70) * 1.15)))
    return 1.0


def clip_coef(norm: float, max_norm: float) -> float:
    return min(1.0, max_norm / (norm + 1e-6))


def adamw_update(p, g, m, v, lr, wd, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """In place on fp32 tensors; the order of operations of torch/optim/adamw.py::_single_tensor_adamw."""
    p.mul_(1 - lr * wd)
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


class StepOracle:
    """Stateful restatement: params/grads are lists of fp32 CPU tensors, groups = [(indices, lr, wd)]."""

    def __init__(self, params: List[torch.Tensor], groups, betas=(0.9, 0.999), eps=1e-8):
        self.params = params
        self.groups = groups
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.betas, self.eps = betas, eps
        self.t = 0
        self.history: List[float] = []

    def step(self, grads: List[torch.Tensor], adaptive: bool, record: bool, fixed_max_norm: float = 0.5) -> Dict[str, float]:
        bad = sanitize_grads(grads)
        norm = total_norm(grads)
        if adaptive:
            if record:
                self.history.append(norm)
            mx = adaptive_max_norm(self.history)
        else:
            mx = fixed_max_norm
        coef = clip_coef(norm, mx)
        self.t += 1
        for idx, lr, wd in self.groups:
            for i in idx:
                adamw_update(self.params[i], grads[i] * coef, self.m[i], self.v[i], lr, wd, self.t, self.betas[0], self.betas[1],
                             self.eps)
        return dict(grad_norm=norm, max_norm=mx, clip_coef=coef, non_finite=bad)


def warmup_cosine(total_epochs: int, warmup_epochs: int, start_factor=0.01, min_factor=0.01):
    def lmbda(epoch):
        if epoch < warmup_epochs:
            return start_factor + (1.0 - start_factor) * (epoch + 1) / max(1, warmup_epochs)
        T = max(1, total_epochs - warmup_epochs)
        t = max(0, epoch - warmup_epochs)
        return min_factor + (1.0 - min_factor) * 0.5 * (1.0 + math.cos(math.pi * t / T))
    return lmbda
