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


# ---------------------------------------------------------------------------------------------------------------------
# N optimizer steps of the reference's loop (train.py:969-1045, the non-AMP branch the reference runs: use_amp=False,
# train.py:1477-1479) on the CPU oracle: autograd through reid_oracle.forward / compute_loss, _sanitize_grads, the adaptive
# clip rule fed by the total norm, torch.nn.utils.clip_grad_norm_, torch.optim.AdamW.step -- the library calls the reference
# makes, on the reference's own parameter groups (tests/golden/learnable_params.json).  Used by tests/test_trajectory_gpu.py to
# hold K steps of the HIP step driver against K steps of fp32 autograd.

def ste_merged_linear(round_fn):
    """A replacement for reid_oracle.mer_linear that evaluates MERLinear the way the HIP path does: ONE product with the merged
    weight round_fn(W + (alpha/r) B A), while the gradient reaches A and B through the unrounded sum (straight-through) -- the
    adapter gradients of the HIP path are formed from the unquantised function (csrc/lora.hip).  round_fn = identity restates
    mer_linear exactly (up to fp32 summation order); round_fn = bf16 / f16 rounding isolates the effect of the merged operand's
    rounding on a training trajectory from every other 16-bit rounding of the HIP path."""
    def mer_linear(x, state, prefix, modality, scaling):
        W = state[prefix + '.shared_linear.weight']
        a = state[f'{prefix}.loras.{modality}.lora_A.weight']
        bm = state[f'{prefix}.loras.{modality}.lora_B.weight']
        wsum = W + scaling * (bm @ a)
        weff = wsum + (round_fn(wsum.detach()) - wsum.detach())
        return x @ weff.t() + state[prefix + '.shared_linear.bias']
    return mer_linear


ROUND_FN = {None: None, 'exact': lambda w: w, 'bf16': lambda w: w.to(torch.bfloat16).float(), 'f16': lambda w: w.half().float()}


class TrajectoryOracle:
    """state: reference-keyed fp32 tensors; groups: [{'name', 'lr', 'params': [keys]}] (the reference's inventory after its freeze
    rule); every listed key trains, everything else is constant.  ``merged`` in ROUND_FN: None = MERLinear as the reference
    writes it; 'exact' / 'bf16' / 'f16' = the merged-weight form with that rounding of the merged operand."""

    def __init__(self, state: Dict[str, torch.Tensor], arch, groups, *, weight_decay=1e-4, adaptive_clip=True, contrastive_weight=0.1,
                 tau=0.2, ce_weight=1.0, use_sdm=True, merged=None, lr_scale=1.0, accum_steps=1, norm_every=200):
        self.arch = arch
        self.state = {k: v.detach().clone().float() if torch.is_tensor(v) and v.dtype.is_floating_point else v for k, v in state.items()}
        self.groups = []
        for g in groups:
            ps = []
            for k in g['params']:
                if k not in self.state:                      # tensors the forward never reads (dead keys of the reference: no gradient,
                    continue                                 # AdamW skips them there too)
                self.state[k].requires_grad_(True)
                ps.append(self.state[k])
            if ps:
                sc = lr_scale.get(g['name'], 1.0) if isinstance(lr_scale, dict) else lr_scale      # {group name: factor} or one factor
                self.groups.append(dict(params=ps, lr=g['lr'] * sc, name=g['name']))
        self.keys = [k for g in groups for k in g['params'] if k in self.state]
        self.opt = torch.optim.AdamW(self.groups, weight_decay=weight_decay, foreach=False)
        self.adaptive, self.history = adaptive_clip, []
        self.loss_kw = dict(ce_weight=ce_weight, contrastive_weight=contrastive_weight, tau=tau, use_sdm=use_sdm)
        self.merged = merged
        self.accum_steps, self.norm_every = max(1, accum_steps), norm_every
        self.batch_idx = 0
        self.norms: List[float] = []

    def step(self, images, tokens, masks, labels) -> Dict[str, float]:
        from . import reid_oracle as O
        bi = self.batch_idx
        if bi % self.accum_steps == 0:
            self.opt.zero_grad(set_to_none=True)
        keep = O.mer_linear
        if ROUND_FN[self.merged] is not None:
            O.mer_linear = ste_merged_linear(ROUND_FN[self.merged])
        try:
            out = O.forward(self.state, self.arch, images, tokens, masks, True)
            L = O.compute_loss(out, labels, **self.loss_kw)
            (L['total_loss'] / self.accum_steps).backward()
        finally:
            O.mer_linear = keep
        if (bi + 1) % self.accum_steps == 0:
            ps = [self.state[k] for k in self.keys]
            gr = [p.grad for p in ps if p.grad is not None]
            sanitize_grads(gr)
            n = total_norm(gr)
            self.norms.append(n)
            if self.adaptive:
                if bi % self.norm_every == 0:
                    self.history.append(n)
                mx = adaptive_max_norm(self.history)
            else:
                mx = 0.5
            torch.nn.utils.clip_grad_norm_(ps, max_norm=float(mx))
            self.opt.step()
        self.batch_idx += 1
        return {k: (float(v.detach()) if torch.is_tensor(v) else v) for k, v in L.items()}
