"""CPU: host-side logic of the package (no GPU): seeded weights, tokenizer, synthetic batches, LoRA arena layout,
config surface, oracle edge cases."""
import os

import pytest
import torch

from prcv2025reid_amd.config import TrainingConfig, arch_of
from prcv2025reid_amd.engine import LoraLayout
from prcv2025reid_amd.synthetic import synthetic_batch
from prcv2025reid_amd.tokenizer import HashTokenizer
from prcv2025reid_amd.weights import param_spec, seeded_tensor, seeded_state, is_dead_key, fingerprint
from oracle import reid_oracle as O


def tiny_cfg(**kw):
    return TrainingConfig(device='cpu', vision_hidden_dim=128, vision_layers=2, vision_heads=2, vision_mlp_dim=256,
                          text_layers=1, text_mlp_dim=1024, text_vocab=1024, text_eos_id=1023, text_bos_id=1022, **kw)


def test_seeded_tensors_do_not_depend_on_order_or_other_keys():
    a = seeded_tensor('clip_encoder.vision_proj.weight', (512, 768), 3)
    b = seeded_tensor('clip_encoder.vision_proj.weight', (512, 768), 3)
    assert torch.equal(a, b) and not torch.equal(a, seeded_tensor('clip_encoder.text_proj.weight', (512, 768), 3))
    assert float(seeded_tensor('bn_neck.bn.running_var', (512,), 0).min()) >= 1.0
    assert float(seeded_tensor('x.lora_B.weight', (8, 4), 0).abs().max()) > 0          # LoRA path is never vacuous


def test_param_spec_matches_reference_key_count():
    arch = arch_of(TrainingConfig(mer_lora_rank=8))
    spec = param_spec(arch, 16)
    assert len([k for k in spec if '.loras.' in k]) == 12 * 6 * 4 * 2
    assert spec['clip_encoder.patch_embeds.nir.proj.weight'] == (768, 1, 16, 16)
    assert spec['clip_encoder.vision_layers.3.mlp.fc2.loras.cp.lora_A.weight'] == (8, 3072)
    assert is_dead_key('clip_encoder.clip_model.vision_model.encoder.layers.0.mlp.fc1.weight')
    assert is_dead_key('clip_encoder.patch_embeds.sk.channel_adapter.weight') and not is_dead_key('bn_neck.bn.weight')


def test_lora_layout_is_a_partition():
    arch = arch_of(TrainingConfig(mer_lora_rank=8))
    lay = LoraLayout(arch)
    assert lay.Rp == 32 and lay.nmod == 4
    spans = []
    for e in lay.ent.values():
        for key in ('A', 'B'):
            o, shp = e[key]
            spans.append((o, o + shp[0] * shp[1]))
    spans.sort()
    assert spans[0][0] == 0 and spans[-1][1] == lay.size
    assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
    t = lay.table()
    assert t.shape == (2 * 12 * 4, 5) and int(t[:, 3:].max()) < lay.pack_size
    assert LoraLayout(arch_of(TrainingConfig(mer_lora_rank=4))).Rp == 32
    assert LoraLayout(arch_of(TrainingConfig(mer_lora_rank=16))).Rp == 64


def test_hash_tokenizer_layout():
    tok = HashTokenizer()
    out = tok(['a b c', '', 'x ' * 100], padding=True, truncation=True, max_length=77)
    ids, am = out['input_ids'], out['attention_mask']
    assert ids.shape == (3, 77) and ids.dtype == torch.int64
    assert ids[0, 0] == 49406 and ids[0, 4] == 49407 and int(am[0].sum()) == 5
    assert ids[1, :2].tolist() == [49406, 49407] and int(am[1].sum()) == 2          # empty string -> BOS EOS
    assert int(am[2].sum()) == 77 and ids[2, 76] == 49407                            # truncated, EOS kept
    assert (ids[0, 5:] == 49407).all()                                               # pad id == EOS id


def test_synthetic_batch_follows_collate_layout():
    arch = arch_of(TrainingConfig())
    b = synthetic_batch(4, 2, arch, seed=1, mask_drop=0.5, num_classes=16)
    assert b['person_id'].tolist() == [0, 0, 1, 1, 2, 2, 3, 3]
    assert set(b['images']) == {'vis', 'nir', 'sk', 'cp'} and b['images']['nir'].shape == (8, 3, 224, 224)
    for m in ('nir', 'sk', 'cp'):
        dead = b['modality_mask'][m] == 0
        assert float(b['images'][m][dead].abs().max() if dead.any() else 0) == 0     # missing image == zeros
    assert all((t == '') == (float(mk) == 0) for t, mk in zip(b['texts'], b['modality_mask']['text']))
    nonvis = torch.stack([b['modality_mask'][m] for m in ('nir', 'sk', 'cp', 'text')], 1)
    assert bool((nonvis.sum(1) >= 1).all()) and bool((b['modality_mask']['vis'] == 1).all())


def test_config_surface_has_the_fields_the_reference_model_reads():
    c = TrainingConfig()
    for name in ('device', 'modalities', 'fusion_dim', 'vision_hidden_dim', 'clip_model_name', 'mer_lora_rank',
                 'mer_lora_alpha', 'drop_path', 'freeze_text_backbone', 'sdm_semantic_dim', 'sdm_num_heads',
                 'fusion_num_heads', 'fusion_mlp_ratio', 'fusion_dropout', 'sdm_temperature', 'ce_weight',
                 'contrastive_weight', 'dropout_rate', 'modality_dropout', 'min_modalities',
                 'modality_dropout_warmup_epochs', 'sdm_weight_warmup_epochs', 'base_learning_rate',
                 'mer_learning_rate', 'tokenizer_learning_rate', 'fusion_learning_rate'):
        assert hasattr(c, name), name
    assert c.mer_lora_rank == 4 and c.contrastive_weight == 0.0 and c.sdm_temperature == 0.2 and c.drop_path == 0.15


@pytest.mark.container
def test_config_defaults_equal_reference_text():
    ref = '/root/reference/configs/config.py'
    if not os.path.exists(ref):
        pytest.skip('reference not mounted')
    import re
    src = open(ref, encoding='utf-8').read()
    ours = TrainingConfig()
    n = 0
    for name, typ, val in re.findall(r'^    (\w+): (int|float|bool|str) = ([^#\n]+)', src, flags=re.M):
        assert hasattr(ours, name), name
        assert repr(getattr(ours, name)) == repr(eval(val.strip())), name
        n += 1
    assert n > 80


# ---- oracle edge cases the reference handles explicitly -------------------------------------------------------
def test_oracle_no_masks_means_no_vision_encoded():
    """models/model.py:367,386-389: with modality_masks=None every vision feature is the null token, mask 0."""
    cfg = tiny_cfg(mer_lora_rank=4)
    arch = arch_of(cfg)
    state = seeded_state(arch, 4, 1)
    b = synthetic_batch(2, 2, arch, seed=2, num_classes=4)
    tok = HashTokenizer(1024, 1022, 1023)(b['texts'])
    out = O.forward(state, arch, {'vis': b['images']['vis']}, tok, None, False)
    assert torch.equal(out['raw_modality_features']['vis'], state['null_tokens.vis'].expand(4, -1))
    assert float(out['feature_masks']['vis'].sum()) == 0 and float(out['feature_masks']['text'].sum()) == 4


def test_oracle_fusion_all_masked_row_and_single_modality():
    cfg = tiny_cfg()
    arch = arch_of(cfg)
    state = seeded_state(arch, None, 3)
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(5, 512, generator=g) for _ in range(3)]
    masks = [torch.tensor([1., 1, 0, 1, 1]), torch.tensor([1., 0, 0, 1, 0]), torch.tensor([0., 1, 0, 1, 1])]
    y = O.feature_fusion(feats, masks, state, 8)
    assert torch.isfinite(y).all() and float(y[2].abs().max()) == 0.0       # all-masked row: masked mean of nothing
    assert torch.equal(O.feature_fusion(feats[:1], masks[:1], state, 8), feats[0])


def test_oracle_loss_skips_invalid_rows_and_missing_pairs():
    logits = torch.randn(6, 5, generator=torch.Generator().manual_seed(1))
    labels = torch.tensor([0, 1, 7, -1, 2, 3])
    fm = {'vis': torch.tensor([1., 1, 1, 1, 0, 1]), 'nir': torch.tensor([0., 0, 0, 0, 0, 0])}
    raw = {'vis': torch.randn(6, 512), 'nir': torch.randn(6, 512)}
    L = O.compute_loss({'logits': logits, 'feature_masks': fm, 'raw_modality_features': raw}, labels)
    assert L['ce_valid_cnt'] == 3                      # rows 2,3 bad label; row 4 no valid modality
    assert float(L['sdm_loss']) == 0.0                 # nir has no valid row -> no pair
    ok = torch.tensor([True, True, False, False, False, True])
    assert abs(float(L['ce_loss']) - float(torch.nn.functional.cross_entropy(logits[ok], labels[ok], label_smoothing=0.1))) < 1e-6


def test_oracle_rank_ties_break_by_index():
    G = torch.nn.functional.normalize(torch.randn(50, 16, generator=torch.Generator().manual_seed(2)), dim=1)
    G[7] = G[3]; G[20] = G[3]
    idx, sc = O.topk_ranklist(G[3:4], G, 4)
    assert idx[0, :3].tolist() == [3, 7, 20]
    ex = torch.zeros(1, 50, dtype=torch.bool); ex[0, 3] = True
    idx2, _ = O.topk_ranklist(G[3:4], G, 2, exclude=ex)
    assert idx2[0].tolist() == [7, 20]


def test_cmc_from_topk_equals_bruteforce_definition():
    """CMC@k over the queries that have a positive in the gallery (eval_mm_protocol.py:425-441); the histogram lookup must select the
    same queries as the [Nq, Ng] comparison it replaced."""
    import torch
    from prcv2025reid_amd.retrieval import cmc_from_topk
    g = torch.Generator().manual_seed(3)
    Nq, Ng, k = 200, 3000, 10
    g_pids = torch.randint(5, 400, (Ng,), generator=g)
    q_pids = torch.randint(0, 450, (Nq,), generator=g)             # some queries have no positive at all
    sim = torch.randn(Nq, Ng, generator=g)
    topk = sim.topk(k, dim=1).indices.int()
    topk[7, 5:] = -1                                               # a short list
    got = cmc_from_topk(topk, q_pids, g_pids, ks=(1, 5, 10))
    has_pos = (q_pids.view(-1, 1) == g_pids.view(1, -1)).any(dim=1)
    assert 0 < int(has_pos.sum()) < Nq
    for kk in (1, 5, 10):
        hit = torch.zeros(Nq, dtype=torch.bool)
        for i in range(Nq):
            idx = topk[i, :kk]; idx = idx[idx >= 0].long()
            hit[i] = bool((g_pids[idx] == q_pids[i]).any())
        want = float(hit[has_pos].float().mean())
        assert abs(got[f'R@{kk}'] - want) < 1e-7, (kk, got, want)
