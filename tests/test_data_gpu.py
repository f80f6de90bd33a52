"""GPU: sampler -> collate -> DeviceFeeder (pinned staging, copy stream) -> StepDriver on the tiny model."""
import random

import pytest
import torch

from helpers import load_case, case_inputs
from test_data_cpu import make_samples

pytestmark = pytest.mark.gpu


def test_feeder_drives_training_steps():
    from prcv2025reid_amd import data as D
    from prcv2025reid_amd.trainer import FusedAdamW, StepDriver
    from test_model_gpu import build_model
    z, meta = load_case('tiny_train_frozen')
    cfg, arch, state, batch, tokens = case_inputs(meta)
    model = build_model(meta, state, True)
    samples = make_samples(4, n_pid=5)
    for s in samples:                                       # labels must be class indices < num_classes (5)
        s['person_id'] = torch.tensor(int(s['person_id']) - 1)
    sm = D.StrictPKBatchSampler(samples, 3, 2, rng=random.Random(7))

    def limited(n=4):
        for i, b in enumerate(sm):
            if i == n:
                return
            yield b
    feeder = D.DeviceFeeder(samples, limited(), model.tokenizer, 'cuda', depth=2)
    gs = [dict(params=[p for p in g['params'] if p.requires_grad], lr=g['lr'], name=g['name']) for g in model.get_learnable_params()]
    drv = StepDriver(model, FusedAdamW(gs, weight_decay=1e-4))
    drv.start_epoch(2)
    losses = []
    for b in feeder:
        assert b['images']['vis'].is_cuda and b['tokens']['input_ids'].is_cuda and not b['modality_mask']['vis'].is_cuda
        L = drv.step(b['images'], b['tokens'], b['modality_mask'], b['person_id'])
        losses.append(float(L['total_loss'].detach()))
    assert len(losses) == 4 and all(x == x and abs(x) < 1e4 for x in losses)
    assert drv.opt.step_count == 4
