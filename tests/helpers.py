"""Shared test helpers: rebuild golden cases from their seeds."""
import os

import numpy as np
import torch

from prcv2025reid_amd.config import TrainingConfig, arch_of
from prcv2025reid_amd.synthetic import synthetic_batch
from prcv2025reid_amd.tokenizer import HashTokenizer
from prcv2025reid_amd.weights import seeded_state, fingerprint

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = {k[5:]: float(z[k]) for k in z.files if k.startswith('meta.')}
    return z, meta


def case_config(meta, device='cpu'):
    return TrainingConfig(
        device=device, mer_lora_rank=int(meta['rank']), mer_lora_alpha=meta['alpha'],
        contrastive_weight=meta['contrastive_weight'], ce_weight=meta['ce_weight'],
        sdm_temperature=meta['tau'], vision_hidden_dim=int(meta['vision_hidden_dim']),
        vision_layers=int(meta['vision_layers']), vision_heads=int(meta['vision_heads']),
        vision_mlp_dim=int(meta['vision_mlp_dim']), text_layers=int(meta['text_layers']),
        text_mlp_dim=int(meta['text_mlp_dim']), text_vocab=int(meta['text_vocab']),
        text_eos_id=int(meta['text_eos_id']), text_bos_id=int(meta['text_bos_id']),
        drop_path=0.0, modality_dropout=0.0, dropout_rate=0.0, fusion_dropout=0.0, sdm_dropout=0.0)


def case_inputs(meta):
    cfg = case_config(meta)
    arch = arch_of(cfg)
    C = int(meta['num_classes'])
    state = seeded_state(arch, C, int(meta['wseed']))
    batch = synthetic_batch(int(meta['P']), int(meta['K']), arch, seed=int(meta['dseed']),
                            mask_drop=meta['mask_drop'], num_classes=C)
    tok = HashTokenizer(arch['text_vocab'], arch['text_bos_id'], arch['text_eos_id'], arch['text_max_len'])
    tokens = tok(batch['texts'], padding=True, truncation=True, max_length=77)
    return cfg, arch, state, batch, tokens


def check_fingerprint(z, state):
    got = fingerprint(state)
    want = float(z['weights_fingerprint'])
    assert abs(got - want) <= 1e-9 * abs(want), (
        f'seeded weights differ from the ones the golden file was made with ({got} vs {want}): '
        'torch RNG drift -- regenerate tests/golden with make_golden.py')


def maxdiff(a, b):
    a = torch.as_tensor(np.asarray(a)).double(); b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max())
