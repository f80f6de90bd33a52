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


def edge_inputs(batch, variant):
    """(images, texts, masks) of the reference call for an edge-case ``variant`` (model.py:367,386-389 no masks; :125-126,
    479-480 single modality; :141-149 all-masked row; :417-418 text default mask).  Shared with the tests."""
    images, texts, masks = batch['images'], batch['texts'], batch['modality_mask']
    if variant == 'nomask':
        return images, texts, None
    if variant == 'single':
        return {'vis': images['vis']}, None, {'vis': masks['vis']}
    if variant == 'textonly':
        return None, texts, None
    if variant == 'novis03':                      # rows 0 and 3 have no RGB image (their other modalities stay)
        masks = {m: t.clone() for m, t in masks.items()}
        images = {m: t.clone() for m, t in images.items()}
        for i in (0, 3):
            masks['vis'][i] = 0.0; images['vis'][i] = 0.0
            masks['nir'][i] = 1.0
        return images, texts, masks
    if variant == 'deadrow':                      # sample 1 has no valid modality at all
        masks = {m: t.clone() for m, t in masks.items()}
        images = {m: t.clone() for m, t in images.items()}
        texts = list(texts)
        for m in masks:
            masks[m][1] = 0.0
        for m in images:
            images[m][1] = 0.0
        texts[1] = ''
        return images, texts, masks
    return images, texts, masks



# fixed modality-dropout draws of the reference-generated fixtures: name -> (forced torch.rand(1) values in the order
# nir, sk, cp, text -- keep iff value > p = 0.5 --, mask_drop of the batch, input variant); epoch 5 > warm-up 3
MODDROP_CASES = {
    'tiny_moddrop_a': ((0.1, 0.9, 0.2, 0.8), 0.0, None),        # nir and cp dropped
    'tiny_moddrop_b': ((0.1, 0.2, 0.3, 0.4), 0.0, None),        # everything but vis dropped -> unfused vis feature
    'tiny_moddrop_c': ((0.1, 0.2, 0.3, 0.4), 0.3, 'novis03'),   # would leave rows 0 and 3 empty -> the draw is cancelled
    'tiny_moddrop_d': ((0.9, 0.2, 0.8, 0.1), 0.3, None),        # sk and text dropped from a 30 % masked batch
}


def check_fingerprint(z, state):
    got = fingerprint(state)
    want = float(z['weights_fingerprint'])
    assert abs(got - want) <= 1e-9 * abs(want), (
        f'seeded weights differ from the ones the golden file was made with ({got} vs {want}): '
        'torch RNG drift -- regenerate tests/golden with make_golden.py')


def maxdiff(a, b):
    a = torch.as_tensor(np.asarray(a)).double(); b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max())


def reference_trainable_groups(tag='tiny_frozen'):
    """The reference's optimiser groups after train.py:1418-1458 -- get_learnable_params() filtered to requires_grad -- from the
    inventory the reference itself produced (tests/golden/learnable_params.json: names, learning rates, trainable flags)."""
    import json
    groups = json.load(open(os.path.join(GOLDEN, 'learnable_params.json')))[tag]
    out = []
    for g in groups:
        keys = [k for k, t in zip(g['params'], g['trainable']) if t]
        if keys:
            out.append(dict(name=g['name'], lr=g['lr'], params=keys))
    return out
