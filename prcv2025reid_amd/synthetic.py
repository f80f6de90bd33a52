"""Seeded synthetic P x K batches in the layout of the reference's collate output.

Layout follows ``compatible_collate_fn`` (datasets/dataset.py:1467-1606, SURVEY.md
section 8b): ``person_id`` int64[B]; ``images`` {vis,nir,sk,cp: fp32[B,3,H,W]} --
three channels for every modality, zeros where the sample lacks it;
``modality_mask`` {vis,nir,sk,cp,text: fp32[B] in {0,1}}; ``text_description``
List[str] ('' where missing).  Data generation follows SURVEY.md section 8(d):
``randn`` images, labels ``arange(P).repeat_interleave(K)``, random-token text.
"""
from typing import Dict, Optional, Sequence

import torch

VMODS = ('vis', 'nir', 'sk', 'cp')


def synthetic_batch(P: int, K: int, arch: dict, seed: int = 0, mask_drop: float = 0.0,
                    num_classes: Optional[int] = None, label_offset: int = 0,
                    drop_vis_rows: Sequence[int] = (), min_words: int = 3, max_words: int = 20,
                    device: str = 'cpu') -> Dict:
    g = torch.Generator(device='cpu').manual_seed(int(seed))
    B = P * K
    H = arch['image_size']
    labels = torch.arange(P).repeat_interleave(K) + label_offset
    if num_classes is not None:
        labels = labels % num_classes
    images = {m: torch.randn(B, 3, H, H, generator=g) for m in VMODS}
    mask = {m: torch.ones(B) for m in VMODS + ('text',)}
    if mask_drop > 0:
        for m in ('nir', 'sk', 'cp', 'text'):
            mask[m] = (torch.rand(B, generator=g) >= mask_drop).float()
        # at least one non-vis modality per sample (BASELINE config 5 / SURVEY 8d)
        nonvis = torch.stack([mask[m] for m in ('nir', 'sk', 'cp', 'text')], 1)
        for i in (nonvis.sum(1) == 0).nonzero().flatten().tolist():
            mask['nir'][i] = 1.0
    for i in drop_vis_rows:
        mask['vis'][i] = 0.0
    for m in VMODS:
        images[m] = images[m] * mask[m].view(B, 1, 1, 1)
    n_words = torch.randint(min_words, max_words + 1, (B,), generator=g)
    texts = []
    for i in range(B):
        if mask['text'][i] > 0:
            ws = torch.randint(0, 1 << 20, (int(n_words[i]),), generator=g).tolist()
            texts.append(' '.join(f'w{w}' for w in ws))
        else:
            texts.append('')
    batch = {'person_id': labels.to(torch.int64), 'images': images, 'modality_mask': mask,
             'text_description': texts, 'texts': texts}
    if device != 'cpu':
        batch['person_id'] = batch['person_id'].to(device)
        batch['images'] = {m: t.to(device) for m, t in images.items()}
        batch['modality_mask'] = {m: t.to(device) for m, t in mask.items()}
    return batch
