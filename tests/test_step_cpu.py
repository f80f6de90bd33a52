"""CPU: the step oracle against the library calls the reference makes (torch.optim.AdamW, clip_grad_norm_, LambdaLR)."""
import math

import numpy as np
import torch

from oracle import step_oracle as so
from prcv2025reid_amd import trainer


def _setup(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(1000,), (33, 7), (5,), (64, 16)]
    params = [torch.randn(*s, generator=g) for s in shapes]
    return params, g


def test_oracle_matches_torch_adamw_and_clip():
    params, g = _setup()
    ref = [torch.nn.Parameter(p.clone()) for p in params]
    groups = [([0, 1], 5e-5, 1e-4), ([2], 3e-3, 1e-4), ([3], 5e-5, 0.0)]
    opt = torch.optim.AdamW([dict(params=[ref[i] for i in idx], lr=lr, weight_decay=wd) for idx, lr, wd in groups], foreach=False)
    orc = so.StepOracle([p.clone() for p in params], groups)
    for step in range(6):
        grads = [torch.randn(p.shape, generator=g) * (10.0 if step % 2 else 0.01) for p in params]
        if step == 3:
            grads[0][5] = float('nan'); grads[1][2, 3] = float('inf')
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        for r in ref:                                        # _sanitize_grads, train.py:85-96
            bad = ~torch.isfinite(r.grad); r.grad[bad] = 0.0
        torch.nn.utils.clip_grad_norm_(ref, max_norm=0.5)
        opt.step()
        orc.step([gr.clone() for gr in grads], adaptive=False, record=False, fixed_max_norm=0.5)
        for r, o in zip(ref, orc.params):
            assert torch.allclose(r.detach(), o, rtol=2e-6, atol=1e-7)


def test_adaptive_rule():
    assert so.adaptive_max_norm([1.0] * 10) == 1.0                      # needs MORE than ten
    h = [float(i) for i in range(1, 12)]                                # last ten: 2..11 -> p70 = 8.3
    assert abs(so.adaptive_max_norm(h) - 3.0) < 1e-12                   # 8.3 * 1.15 clamps to 3
    h = [0.1 * i for i in range(1, 12)]                                 # p70 = 0.83 -> 0.9545
    assert abs(so.adaptive_max_norm(h) - 0.83 * 1.15) < 1e-9
    assert so.adaptive_max_norm([0.01] * 11) == 0.5


def test_lambda_matches_lambdalr():
    lm = so.warmup_cosine(20, 5)
    lm2 = trainer.warmup_cosine_lambda(20, 5)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lm)
    for e in range(25):
        assert abs(opt.param_groups[0]['lr'] - lm(e)) < 1e-12 and lm(e) == lm2(e)
        opt.step(); sch.step()
    assert abs(lm(4) - 1.0) < 1e-12 and abs(lm(20) - 0.01) < 1e-12
