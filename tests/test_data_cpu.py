"""CPU: the P x K sampler and the collate against what the reference's own code produced on the same seeded samples
(tests/golden/pipeline_cases.json, written by make_golden.py --only pipeline from /root/reference/datasets/dataset.py)."""
import json
import os
import random

import pytest
import torch

from helpers import GOLDEN
from prcv2025reid_amd import data as D

CASES = json.load(open(os.path.join(GOLDEN, 'pipeline_cases.json')))


def make_samples(seed, n_pid=9, image_size=224):
    """Same generator as tests/golden/make_golden.py::_pipeline_samples (inputs are re-created from the seed, not shipped)."""
    g = torch.Generator().manual_seed(seed)
    R = random.Random(seed)
    samples = []
    for pid in range(1, n_pid + 1):
        for j in range(R.randint(2, 7)):
            kind = R.choice(['vis', 'vis', 'nir', 'sk', 'cp', 'multi'])
            imgs, mask = {}, {}
            mods = ['vis', 'nir', 'sk', 'cp'] if kind == 'multi' else [kind]
            if pid == 3:
                mods = ['vis']
            for m in ['vis', 'nir', 'sk', 'cp']:
                if m in mods:
                    imgs[m] = torch.randn(3, image_size, image_size, generator=g)
                    mask[m] = 1.0
                elif R.random() < 0.3:
                    imgs[m] = torch.zeros(3, image_size, image_size)
                    mask[m] = 1.0
                else:
                    mask[m] = 0.0
            if R.random() < 0.15 and 'nir' in imgs:
                mask['nir'] = False
            cap = '' if pid == 3 else R.choice(['a person walking', '  ', 'red coat, black bag', ''])
            s = {'person_id': torch.tensor(pid), 'images': imgs, 'modality_mask': mask,
                 'text_description': [cap] if R.random() < 0.8 else cap}
            if R.random() < 0.5:
                s['modality'] = R.choice(['RGB', 'ir', 'sketch', 'cpencil', 'vis'])
            samples.append(s)
    return samples


@pytest.mark.parametrize('case', CASES['sampler'], ids=lambda c: f"seed{c['seed']}")
def test_sampler_reproduces_reference_draws(case):
    samples = make_samples(case['seed'])
    sm = D.StrictPKBatchSampler(samples, case['P'], case['K'], allow_id_reuse=case['reuse'], rng=random.Random(case['rng_seed']))
    assert sm.paired_ids == case["strong_ids"] and sm.unpaired_ids == case["soft_ids"] and len(sm) == case['len']
    got = []
    for b in sm:
        got.append(b)
        if len(got) >= len(case['batches']):
            break
    assert got == case['batches']
    for b in got:                                    # the structure the SDM loss relies on
        pids = [int(samples[i]['person_id']) for i in b]
        assert len(b) == case['P'] * case['K']
        for p in range(case['P']):
            assert len(set(pids[p * case['K']:(p + 1) * case['K']])) == 1


@pytest.mark.parametrize('case', CASES['collate'], ids=lambda c: f"seed{c['seed']}")
def test_collate_matches_reference(case):
    samples = make_samples(case['seed'])
    b = D.collate([samples[i] for i in case['indices']])
    assert b['person_id'].tolist() == case['person_id']
    assert b['text_description'] == case['text_description']
    assert b['modality'] == case['modality']
    for m, v in case['modality_mask'].items():
        assert b['modality_mask'][m].tolist() == v, m
    for m, shp in case['image_shapes'].items():
        assert list(b['images'][m].shape) == shp
        got = b['images'][m].double().flatten(1).sum(1).tolist()
        assert all(abs(x - y) <= 1e-9 * max(1.0, abs(y)) for x, y in zip(got, case['image_sums'][m])), m


def perturb(samples, seed):
    """Same function as tests/golden/make_golden.py::_pipeline_perturb."""
    R = random.Random(1000 + seed)
    for s in samples:
        if R.random() < 0.2:
            s['mode'] = R.choice(['ir', 'x', 'sketch'])
        if R.random() < 0.2:
            s.pop('modality', None); s['mod'] = R.choice(['RGB', 'cp'])
        if R.random() < 0.1:
            s['images']['text'] = 'hello'
        if R.random() < 0.1:
            s['text_description'] = []
        if R.random() < 0.1:
            s['text_description'] = ['  ']
        if R.random() < 0.1:
            s['modality_mask']['cp'] = 2
        if R.random() < 0.1:
            s['modality_mask'] = {}
    return samples


@pytest.mark.parametrize('case', CASES['variants'], ids=lambda c: f"seed{c['seed']}")
def test_variants_match_reference(case):
    """Alternate field spellings, odd captions, odd K, an index subset, no identity reuse: modality inference per sample, the identity
    lists, every batch the reference completes, and the collate of the first batch."""
    samples = perturb(make_samples(case['seed'], n_pid=7), case['seed'])
    for i, s in enumerate(samples):
        assert [sorted(D.infer_modalities(s, True)), sorted(D.infer_modalities(s, False))] == case['infer'][i], i
    sm = D.StrictPKBatchSampler(samples, case['P'], case['K'], allow_id_reuse=case['reuse'], indices=case['keep'],
                                rng=random.Random(case['rng_seed']))
    assert sm.paired_ids == case['strong_ids'] and sm.unpaired_ids == case['soft_ids'] and len(sm) == case['len']
    it = iter(sm)
    got = [next(it) for _ in case['batches']]
    assert got == case['batches']
    if not case['reuse']:                                # where the reference starts retrying forever this iterator ends
        rest = list(it)
        assert len(rest) <= 1 and all(len(b) == case['P'] * case['K'] for b in rest)
    if 'collate' in case:
        c = case['collate']
        b = D.collate([samples[i] for i in c['indices']])
        assert b['person_id'].tolist() == c['person_id'] and b['text_description'] == c['text_description'] and b['modality'] == c['modality']
        for m, v in c['modality_mask'].items():
            assert b['modality_mask'][m].tolist() == v, m
        for m, sums in c['image_sums'].items():
            got_s = b['images'][m].double().flatten(1).sum(1).tolist()
            assert all(abs(x - y) <= 1e-9 * max(1.0, abs(y)) for x, y in zip(got_s, sums)), m


def test_sampler_ends_when_identities_run_out():
    """Fewer identities than P and nothing to fill up with: the reference retries the same short batch forever; here the epoch ends."""
    samples = make_samples(0, n_pid=2)
    for reuse in (True, False):
        sm = D.StrictPKBatchSampler([s for s in samples if int(s['person_id']) != 3], 4, 2, allow_id_reuse=reuse, rng=random.Random(0))
        assert list(sm) == []


def test_canon_and_infer():
    assert D.canon_mod(' RGB ') == 'vis' and D.canon_mod('cpencil') == 'cp' and D.canon_mod(None) == '' and D.canon_mod('x') == 'x'
    s = {'modality_mask': {'ir': 1.0, 'vis': 0.0}, 'images': {'sketch': torch.ones(1)}, 'mode': 'colorpencil', 'caption': 'hi'}
    assert D.infer_modalities(s) == {'nir', 'sk', 'cp', 'text'} and D.infer_modalities(s, include_text=False) == {'nir', 'sk', 'cp'}


def test_feeder_cpu_path():
    from prcv2025reid_amd.tokenizer import HashTokenizer
    samples = make_samples(0, n_pid=4, image_size=224)
    sm = D.StrictPKBatchSampler(samples, 2, 2, rng=random.Random(0))
    def limited():
        for i, b in enumerate(sm):
            if i == 3:
                return
            yield b
    f = D.DeviceFeeder(samples, limited(), HashTokenizer(1024, 1022, 1023, 77), 'cpu', depth=2)
    n = 0
    for b in f:
        assert set(b['images']) == {'vis', 'nir', 'sk', 'cp'} and b['tokens']['input_ids'].shape[0] == 4
        assert b['modality_mask']['vis'].shape == (4,)
        n += 1
    assert n == 3
