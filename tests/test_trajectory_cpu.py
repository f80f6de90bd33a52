"""CPU: the trajectory oracle (oracle/step_oracle.py TrajectoryOracle) is what it says -- autograd + torch.optim.AdamW on the
reference's parameter groups -- and its merged-weight form with an identity rounding is the same function as MERLinear."""
import json
import os

import torch

from helpers import GOLDEN, load_case, case_inputs, reference_trainable_groups
from oracle import step_oracle as so


def _setup(zero_b=False):
    z, meta = load_case('tiny_train_frozen')
    cfg, arch, state, batch, tokens = case_inputs(meta)
    if zero_b:
        state = {k: (torch.zeros_like(v) if k.endswith('lora_B.weight') else v) for k, v in state.items()}
    groups = reference_trainable_groups('tiny_frozen')
    kw = dict(contrastive_weight=meta['contrastive_weight'], tau=meta['tau'], ce_weight=meta['ce_weight'])
    return state, arch, batch, tokens, groups, kw


def test_trajectory_oracle_first_loss_is_the_fixture_loss_and_steps_descend():
    z, meta = load_case('tiny_train_frozen')
    state, arch, batch, tokens, groups, kw = _setup()
    t = so.TrajectoryOracle(state, arch, groups, lr_scale={'mer_loras': 50.0}, **kw)
    losses = [t.step(batch['images'], tokens, batch['modality_mask'], batch['person_id'])['total_loss'] for _ in range(4)]
    assert abs(losses[0] - float(z['total_loss'])) <= 5e-5 * abs(float(z['total_loss']))     # step 0 = the reference's own loss
    assert losses[-1] < 0.7 * losses[0]
    moved = [k for k in t.keys if float((t.state[k].detach() - state[k]).abs().max()) > 0]
    assert all(k in moved for k in t.keys if '.loras.' in k) and 'bn_neck.classifier.weight' in moved and len(moved) >= 0.8 * len(t.keys)
    assert all(('loras' in k or 'bn_neck' in k or 'null_tokens' in k) for k in t.keys)      # train.py:1421-1425
    frozen = [k for k in state if k not in t.keys and torch.is_tensor(state[k]) and state[k].dtype.is_floating_point and 'running' not in k and 'num_batches' not in k]
    assert all(torch.equal(t.state[k], state[k].float()) for k in frozen)


def test_merged_form_with_identity_rounding_is_mer_linear():
    state, arch, batch, tokens, groups, kw = _setup()
    a = so.TrajectoryOracle(state, arch, groups, lr_scale={'mer_loras': 50.0}, **kw)
    b = so.TrajectoryOracle(state, arch, groups, lr_scale={'mer_loras': 50.0}, merged='exact', **kw)
    for _ in range(3):
        la = a.step(batch['images'], tokens, batch['modality_mask'], batch['person_id'])
        lb = b.step(batch['images'], tokens, batch['modality_mask'], batch['person_id'])
        assert abs(la['total_loss'] - lb['total_loss']) <= 2e-5
    num = sum(float((a.state[k].detach() - b.state[k].detach()).pow(2).sum()) for k in a.keys if '.loras.' in k)
    den = sum(float((a.state[k].detach() - state[k]).pow(2).sum()) for k in a.keys if '.loras.' in k)
    assert num <= (0.02 ** 2) * den          # Adam's sign-like first steps magnify fp32 summation-order noise on near-zero gradients


def test_reference_init_has_zero_lora_b_and_trains():
    state, arch, batch, tokens, groups, kw = _setup(zero_b=True)
    t = so.TrajectoryOracle(state, arch, groups, **kw)
    for _ in range(2):
        t.step(batch['images'], tokens, batch['modality_mask'], batch['person_id'])
    kb = [k for k in t.keys if k.endswith('lora_B.weight')]
    assert kb and all(float(t.state[k].detach().abs().max()) > 0 for k in kb)
