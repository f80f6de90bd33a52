#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference itself.  BUILD CONTAINER ONLY.

Runs the reference's own modules (``/root/reference/models/*``, imported, never
copied) on seeded synthetic inputs and stores inputs-by-seed + the reference's
outputs, losses and gradients as small ``.npz`` fixtures.  Nothing here travels
to the GPU box except the fixtures; the tests re-create weights and inputs from
the seeds with ``prcv2025reid_amd.weights.seeded_tensor`` / ``synthetic_batch``.

How the reference is made to run offline:
  * ``models.clip_backbone.CLIPModel`` is rebound to an object whose
    ``from_pretrained`` returns a locally constructed random-init
    ``transformers.CLIPModel(CLIPConfig(...))`` -- the hub fetch by NAME
    (clip_backbone.py:170) is impossible offline and is never attempted;
  * ``models.clip_backbone.CLIPTokenizer`` is rebound to this package's
    ``HashTokenizer`` (the BPE vocabulary is a download too);
  * every tensor of the reference model's ``state_dict()`` is then overwritten
    by ``seeded_fill`` so the values do not depend on HF's initialiser;
  * all dropouts / DropPath / modality-dropout are switched off, and
    ``contrastive_weight`` / ``current_epoch`` set so SDM is computed (F7).

Retrieval metrics: ``train.py`` and ``tools/eval_mm_protocol.py`` cannot be
imported (``import torchvision`` -> ModuleNotFoundError), so the functions
``_reid_map`` (train.py:450-479) and ``rank_and_metrics``
(eval_mm_protocol.py:369-469) are pulled out of the source files by ``ast`` and
executed from there, with ``extract_query_feat`` bound to a lookup of precomputed
query features (the tool's model half is broken as shipped).

Usage:  python tests/golden/make_golden.py [--only tiny|full|retrieval]
"""
import argparse
import ast
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from prcv2025reid_amd.config import TrainingConfig, arch_of            # noqa: E402
from prcv2025reid_amd.tokenizer import HashTokenizer                    # noqa: E402
from prcv2025reid_amd.weights import seeded_fill, fingerprint, param_spec  # noqa: E402
from prcv2025reid_amd.synthetic import synthetic_batch                  # noqa: E402
sys.path.insert(0, os.path.join(REPO, 'tests'))
from helpers import edge_inputs, MODDROP_CASES                          # noqa: E402  (the same input recipe the tests use)


RANDOMIZE_HF = False


def build_reference(cfg, num_classes, seed):
    from transformers import CLIPModel, CLIPConfig
    import models.clip_backbone as cb
    a = arch_of(cfg)
    hf_cfg = CLIPConfig(
        vision_config=dict(hidden_size=a['vision_hidden_dim'], intermediate_size=a['vision_mlp_dim'],
                           num_hidden_layers=a['vision_layers'], num_attention_heads=a['vision_heads'],
                           image_size=a['image_size'], patch_size=a['patch_size']),
        text_config=dict(hidden_size=a['text_hidden_dim'], intermediate_size=a['text_mlp_dim'],
                         num_hidden_layers=a['text_layers'], num_attention_heads=a['text_heads'],
                         vocab_size=a['text_vocab'], max_position_embeddings=a['text_max_len'],
                         eos_token_id=a['text_eos_id'], bos_token_id=a['text_bos_id'],
                         pad_token_id=a['text_eos_id']),
        projection_dim=a['fusion_dim'])

    class _LocalCLIP:
        @staticmethod
        def from_pretrained(name):
            m = CLIPModel(hf_cfg)
            if RANDOMIZE_HF:                     # distinct values everywhere: construction-time copies become identifiable
                with torch.no_grad():
                    for p in m.parameters():
                        p.normal_()
            return m

    class _LocalTok:
        @staticmethod
        def from_pretrained(name):
            return HashTokenizer(a['text_vocab'], a['text_bos_id'], a['text_eos_id'], a['text_max_len'])

    cb.CLIPModel = _LocalCLIP
    cb.CLIPTokenizer = _LocalTok
    from models.model import CLIPBasedMultiModalReIDModel
    m = CLIPBasedMultiModalReIDModel(cfg)
    m.set_num_classes(num_classes)
    assert m.clip_encoder.clip_model.text_model.config.eos_token_id == a['text_eos_id']
    seeded_fill(m.state_dict(), seed)
    # regularisers off
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
        if mod.__class__.__name__ == 'DropPath':
            mod.drop_prob = 0.0
    return m


def apply_freeze_rule(model):
    """train.py:1420-1425."""
    for name, p in model.named_parameters():
        p.requires_grad = ('loras' in name or 'feature_mixture' in name or 'bn_neck' in name
                           or 'null_tokens' in name)


GRAD_KEEP = ('null_tokens.', 'bn_neck.',
             'vision_layers.0.attn.q_proj.loras.', 'vision_layers.0.mlp.fc2.loras.',
             'vision_layers.1.attn.v_proj.loras.nir', 'vision_layers.1.mlp.fc1.loras.sk',
             'vision_layers.11.attn.out_proj.loras.', 'vision_layers.11.mlp.fc1.loras.cp')
GRAD_KEEP_FULLTRAIN = GRAD_KEEP + (
    'vision_layers.0.ln1.', 'vision_layers.0.attn.k_proj.shared_linear.bias', 'vision_ln_final.',
    'cls_token', 'patch_embeds.nir.proj.bias', 'feature_fusion.norm1.', 'sdm_module.semantic_proj.1.',
    'text_model.final_layer_norm.', 'text_model.encoder.layers.0.layer_norm1.',
    'text_model.encoder.layers.0.self_attn.q_proj.bias')


class _ForcedRand:
    """Stands in for ``torch.rand`` while the reference's forward runs: ``torch.rand(1)`` (the modality-dropout draw,
    model.py:450) returns the next forced value; every other call goes to the real function."""

    def __init__(self, values):
        self.values = list(values); self.real = torch.rand; self.used = 0

    def __call__(self, *size, **kw):
        if size == (1,) and not kw and self.values:
            self.used += 1
            return torch.tensor([self.values.pop(0)])
        return self.real(*size, **kw)


def run_case(name, cfg, *, num_classes, wseed, dseed, P, K, mask_drop, training, freeze, keep, variant=None, forced=None,
             epoch=None):
    print(f'[{name}] building reference ...', flush=True)
    a = arch_of(cfg)
    model = build_reference(cfg, num_classes, wseed)
    batch = synthetic_batch(P, K, a, seed=dseed, mask_drop=mask_drop, num_classes=num_classes)
    model.train(training)
    if freeze:
        apply_freeze_rule(model)
    model.set_epoch(epoch if epoch is not None else getattr(cfg, 'golden_epoch', 2))
    out = {}
    spec = param_spec(a, num_classes)
    sd = model.state_dict()
    out['weights_fingerprint'] = np.float64(fingerprint({k: sd[k] for k in spec}))
    images, texts, masks = edge_inputs(batch, variant)
    with torch.set_grad_enabled(training):
        if forced is not None:
            fr = _ForcedRand(forced)
            torch.rand = fr
            try:
                o = model(images=images, texts=texts, modality_masks=masks)
            finally:
                torch.rand = fr.real
            out['forced_used'] = np.int64(fr.used)
        else:
            o = model(images=images, texts=texts, modality_masks=masks)
        out['fused_modalities'] = np.array(list(o['modality_features'].keys()))
        for k in ('features', 'bn_features', 'logits'):
            out[k] = o[k].detach().numpy()
        for m, t in o['raw_modality_features'].items():
            out[f'raw.{m}'] = t.detach().numpy()
        for m, t in o['modality_features'].items():
            out[f'sem.{m}'] = t.detach().numpy()
        for m, t in o['feature_masks'].items():
            out[f'fmask.{m}'] = t.detach().numpy()
        if training:
            L = model.compute_loss(o, batch['person_id'])
            for k in ('total_loss', 'ce_loss', 'sdm_loss'):
                out[k] = np.float64(float(L[k]))
            out['ce_valid_cnt'] = np.int64(L['ce_valid_cnt'])
            L['total_loss'].backward()
            sumsq = 0.0
            for n, p in model.named_parameters():
                if p.grad is None:
                    continue
                sumsq += float(p.grad.double().pow(2).sum())
                if any(s in n for s in keep):
                    out[f'grad.{n}'] = p.grad.detach().numpy()
            out['grad_sumsq'] = np.float64(sumsq)
            out['bn_running_mean'] = model.bn_neck.bn.running_mean.numpy()
            out['bn_running_var'] = model.bn_neck.bn.running_var.numpy()
    meta = dict(P=P, K=K, wseed=wseed, dseed=dseed, mask_drop=mask_drop, num_classes=num_classes,
                training=int(training), freeze=int(freeze),
                rank=cfg.mer_lora_rank, alpha=cfg.mer_lora_alpha, tau=cfg.sdm_temperature,
                contrastive_weight=cfg.contrastive_weight, ce_weight=cfg.ce_weight,
                vision_hidden_dim=a['vision_hidden_dim'], vision_layers=a['vision_layers'],
                vision_heads=a['vision_heads'], vision_mlp_dim=a['vision_mlp_dim'],
                text_layers=a['text_layers'], text_mlp_dim=a['text_mlp_dim'], text_vocab=a['text_vocab'],
                text_eos_id=a['text_eos_id'], text_bos_id=a['text_bos_id'])
    for k, v in meta.items():
        out[f'meta.{k}'] = np.float64(v)
    path = os.path.join(HERE, f'{name}.npz')
    np.savez_compressed(path, **out)
    print(f'[{name}] total={out.get("total_loss")} ce={out.get("ce_loss")} sdm={out.get("sdm_loss")} -> {path} '
          f'({os.path.getsize(path) / 1e3:.0f} kB)', flush=True)


def tiny_cfg(rank=4):
    c = TrainingConfig(device='cpu', mer_lora_rank=rank, contrastive_weight=0.1, vision_hidden_dim=128,
                       vision_layers=2, vision_heads=2, vision_mlp_dim=256, text_layers=2, text_mlp_dim=1024,
                       text_vocab=1024, text_eos_id=1023, text_bos_id=1022, drop_path=0.0,
                       modality_dropout=0.0)
    return c


def full_cfg(rank):
    return TrainingConfig(device='cpu', mer_lora_rank=rank, contrastive_weight=0.1, drop_path=0.0,
                          modality_dropout=0.0)


def moddrop_cfg():
    c = tiny_cfg(4)
    c.modality_dropout = 0.5
    c.modality_dropout_warmup_epochs = 3
    return c


def make_edge_cases():
    """Reference outputs for the forward edge cases and for fixed modality-dropout draws (tiny architecture, train mode)."""
    for v in ('nomask', 'single', 'textonly', 'deadrow'):
        run_case(f'tiny_edge_{v}', tiny_cfg(4), num_classes=5, wseed=14, dseed=24, P=3, K=2, mask_drop=0.3,
                 training=True, freeze=True, keep=('null_tokens.', 'bn_neck.'), variant=v)
    run_case('tiny_edge_single_eval', tiny_cfg(4), num_classes=5, wseed=14, dseed=24, P=3, K=2, mask_drop=0.3,
             training=False, freeze=True, keep=(), variant='single')
    # forced draws (keep iff value > 0.5), order nir, sk, cp, text; epoch 5 > warm-up 3
    #   A: drop nir + cp (masks all-on, so nobody is left empty)       B: drop everything but vis -> unfused vis feature
    #   C: 30 % masked batch with vis missing on rows 0 and 3: dropping all non-vis would empty them -> draw cancelled
    for name, (forced, mask_drop, variant) in MODDROP_CASES.items():
        run_case(name, moddrop_cfg(), num_classes=5, wseed=15, dseed=25, P=3, K=2, mask_drop=mask_drop, training=True,
                 freeze=True, keep=('null_tokens.', 'bn_neck.'), forced=list(forced), epoch=5, variant=variant)


def make_param_groups():
    """learnable_params.json: the reference's optimiser groups (models/model.py:661-729, clip_backbone.py:342-371): group
    names, learning rates and member parameter names, as constructed and after train.py's freeze rule."""
    import json
    out = {}
    for tag, cfg, C in (('tiny', tiny_cfg(4), 5), ('full', full_cfg(8), 16)):
        for frozen in (False, True):
            m = build_reference(cfg, C, 0)
            if frozen:
                apply_freeze_rule(m)
            names = {id(p): n for n, p in m.named_parameters()}
            groups = []
            for g in m.get_learnable_params():
                groups.append({'name': g['name'], 'lr': float(g['lr']), 'params': [names[id(p)] for p in g['params']],
                               'trainable': [bool(p.requires_grad) for p in g['params']]})
            out[f'{tag}_{"frozen" if frozen else "built"}'] = groups
            print(f'[param_groups/{tag}/{"frozen" if frozen else "built"}] ' +
                  ', '.join(f"{g['name']}:{len(g['params'])}@{g['lr']}" for g in groups))
    json.dump(out, open(os.path.join(HERE, 'learnable_params.json'), 'w'))


def make_model_cases(which):
    if which in (None, 'tiny'):
        run_case('tiny_train_r16_masked', tiny_cfg(16), num_classes=5, wseed=16, dseed=26, P=3, K=2, mask_drop=0.3,
                 training=True, freeze=True, keep=GRAD_KEEP)
        run_case('tiny_train_frozen', tiny_cfg(4), num_classes=5, wseed=11, dseed=21, P=3, K=2, mask_drop=0.3,
                 training=True, freeze=True, keep=GRAD_KEEP)
        run_case('tiny_train_all', tiny_cfg(8), num_classes=7, wseed=12, dseed=22, P=4, K=2, mask_drop=0.0,
                 training=True, freeze=False, keep=GRAD_KEEP_FULLTRAIN)
        run_case('tiny_eval', tiny_cfg(16), num_classes=5, wseed=13, dseed=23, P=3, K=2, mask_drop=0.3,
                 training=False, freeze=True, keep=())
    if which in (None, 'full'):
        # BASELINE.json config 1: P=4,K=2, r=4, masks all on, one SDM+CE step
        run_case('full_p4k2_r4', full_cfg(4), num_classes=16, wseed=0, dseed=1, P=4, K=2, mask_drop=0.0,
                 training=True, freeze=True, keep=GRAD_KEEP)
        # masks with 30 % dropout, r=8 (BASELINE configs 2/5 flavour), still CPU-sized
        run_case('full_p4k2_r8_masked', full_cfg(8), num_classes=16, wseed=2, dseed=3, P=4, K=2, mask_drop=0.3,
                 training=True, freeze=True, keep=GRAD_KEEP)
        run_case('full_eval_r8', full_cfg(8), num_classes=16, wseed=2, dseed=4, P=4, K=2, mask_drop=0.3,
                 training=False, freeze=True, keep=())
    if which in (None, 'full', 'full16'):
        # BASELINE config 5 flavour at CPU size: r=16 (alpha/r = 1/16, mer_lora.py:27), 30 % mask drop, TRAIN mode
        run_case('full_p4k2_r16_masked', full_cfg(16), num_classes=16, wseed=5, dseed=6, P=4, K=2, mask_drop=0.3,
                 training=True, freeze=True, keep=('null_tokens.', 'bn_neck.', 'vision_layers.0.attn.q_proj.loras.nir',
                                                   'vision_layers.11.mlp.fc1.loras.cp'))


# --------------------------------------------------------------------------- retrieval
def _extract_functions(path, names):
    src = open(path, encoding='utf-8').read()
    tree = ast.parse(src)
    chunks = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []
            chunks.append(ast.get_source_segment(src, node).split('def ', 1)[1])
    return ['def ' + c for c in chunks]


def make_retrieval_cases():
    import torch.nn.functional as F
    ns = {'torch': torch, 'np': np, 'F': F, 'tqdm': (lambda it, **kw: it)}
    import typing
    ns.update({n: getattr(typing, n) for n in ('List', 'Dict', 'Tuple', 'Optional')})
    for code in _extract_functions(os.path.join(REF, 'train.py'), {'_reid_map'}):
        exec(compile(code, 'train.py', 'exec'), ns)
    for code in _extract_functions(os.path.join(REF, 'tools', 'eval_mm_protocol.py'),
                                   {'cosine_sim', 'rank_and_metrics'}):
        ns['FeatureExtractor'] = object
        exec(compile(code, 'eval_mm_protocol.py', 'exec'), ns)
    ns['extract_query_feat'] = lambda q, extractor, weight_cfg: q['feat']

    g = torch.Generator().manual_seed(5)
    Nq, Ng, D, n_id = 48, 1500, 512, 40
    Q = F.normalize(torch.randn(Nq, D, generator=g), dim=1)
    G = F.normalize(torch.randn(Ng, D, generator=g), dim=1)
    q_pid = torch.randint(0, n_id + 5, (Nq,), generator=g)       # a few ids absent from the gallery
    g_pid = torch.randint(0, n_id, (Ng,), generator=g)
    # make positives rank higher than chance so AP is not degenerate
    for i in range(Nq):
        pos = (g_pid == q_pid[i]).nonzero().flatten()
        if len(pos):
            Q[i] = F.normalize(Q[i] + 0.8 * G[pos[: max(1, len(pos) // 2)]].mean(0), dim=0)
    sim = Q @ G.t()
    mAP, top1 = ns['_reid_map'](sim, q_pid, g_pid)
    # rank_and_metrics with same-img masking: every query shares an img_id with 0-2 gallery rows
    g_img = [f'g{i}' for i in range(Ng)]
    queries = []
    q_img = []
    for i in range(Nq):
        own = [g_img[j] for j in torch.randint(0, Ng, (int(i % 3),), generator=g).tolist()]
        pos = (g_pid == q_pid[i]).nonzero().flatten().tolist()
        if pos and i % 2 == 0:
            own.append(g_img[pos[0]])       # mask a true positive: exercises "& mask"
        q_img.append(own)
        queries.append({'pid': int(q_pid[i]), 'feat': Q[i], 'modalities': ['x'],
                        'samples': {f's{k}': {'img_id': iid} for k, iid in enumerate(own)}})
    meta = [{'pid': int(g_pid[j]), 'img_id': g_img[j]} for j in range(Ng)]
    res = ns['rank_and_metrics'](queries, G, meta, None, {}, ignore_same_img=True)
    res_nomask = ns['rank_and_metrics'](queries, G, meta, None, {}, ignore_same_img=False)
    out = dict(Q=Q.numpy(), G=G.numpy(), q_pid=q_pid.numpy(), g_pid=g_pid.numpy(),
               reid_map=np.float64(mAP), reid_top1=np.float64(top1),
               q_img=np.array(['|'.join(x) for x in q_img]),
               rm_mAP=np.float64(res['mAP']), rm_r1=np.float64(res['R@1']), rm_r5=np.float64(res['R@5']),
               rm_r10=np.float64(res['R@10']), rm_n=np.int64(res['num_queries']),
               rn_mAP=np.float64(res_nomask['mAP']), rn_r1=np.float64(res_nomask['R@1']),
               rn_r5=np.float64(res_nomask['R@5']), rn_r10=np.float64(res_nomask['R@10']),
               rn_n=np.int64(res_nomask['num_queries']))
    path = os.path.join(HERE, 'retrieval_metrics.npz')
    np.savez_compressed(path, **out)
    print(f'[retrieval] _reid_map mAP={mAP:.6f} top1={top1:.4f}; rank_and_metrics={res} -> {path}')

    # SDM known answer straight from the reference function (models/sdm_loss.py:153-167 recipe)
    from models.sdm_loss import sdm_loss_stable
    torch.manual_seed(0)
    qry = torch.randn(16, 768); gal = torch.randn(48, 768)
    ql = torch.randint(0, 10, (16,)); gl = torch.randint(0, 10, (48,))
    y = (ql.view(-1, 1) == gl.view(1, -1)).float()
    v = float(sdm_loss_stable(qry, gal, y, tau=0.2))
    # second point: tau outside the clamp, rows without positives
    y2 = y.clone(); y2[:5] = 0
    v2 = float(sdm_loss_stable(qry, gal, y2, tau=0.05))
    np.savez_compressed(os.path.join(HERE, 'sdm_known.npz'), quick_check=np.float64(v), tau005_rows5zero=np.float64(v2))
    print(f'[sdm] quick_check={v!r} second={v2!r}')


# --------------------------------------------------------------------------- input pipeline
def _pipeline_samples(seed, n_pid=9, image_size=224):
    """Synthetic sample dictionaries in the reference dataset's format, with every irregularity the collate handles."""
    import random as pyrandom
    g = torch.Generator().manual_seed(seed)
    R = pyrandom.Random(seed)
    samples = []
    for pid in range(1, n_pid + 1):
        for j in range(R.randint(2, 7)):
            kind = R.choice(['vis', 'vis', 'nir', 'sk', 'cp', 'multi'])
            imgs, mask = {}, {}
            mods = ['vis', 'nir', 'sk', 'cp'] if kind == 'multi' else [kind]
            if pid == 3:
                mods = ['vis']                      # a soft identity: vis only, no caption
            for m in ['vis', 'nir', 'sk', 'cp']:
                if m in mods:
                    imgs[m] = torch.randn(3, image_size, image_size, generator=g)
                    mask[m] = 1.0
                elif R.random() < 0.3:
                    imgs[m] = torch.zeros(3, image_size, image_size)      # zero placeholder that claims to be present
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


def _pipeline_perturb(samples, seed):
    """Alternate spellings of the sample fields (same function in tests/test_data_cpu.py)."""
    import random as pyrandom
    R = pyrandom.Random(1000 + seed)
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


def make_pipeline_cases():
    """pipeline_cases.json: the reference's own sampler and collate (datasets/dataset.py, pulled out by ``ast`` because the
    module imports torchvision) run on seeded synthetic samples: index lists of the strict P x K sampler under a seeded
    ``random``, and the collate's masks / primary modalities / image checksums."""
    import json
    import random as pyrandom
    path = os.path.join(REF, 'datasets', 'dataset.py')
    src = open(path, encoding='utf-8').read()
    tree = ast.parse(src)
    want_f = {'canon_mod', '_truthy', 'infer_modalities_of_sample', 'compatible_collate_fn'}
    want_a = {'CANON_DS', 'IMG_MODALITIES', 'ALL_MODALITIES'}
    chunks = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in want_f:
            chunks.append(ast.get_source_segment(src, node))
        elif isinstance(node, ast.ClassDef) and node.name == 'ModalAwarePKBatchSampler_Strict':
            chunks.append(ast.get_source_segment(src, node))
        elif isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in want_a for t in node.targets):
            chunks.append(ast.get_source_segment(src, node))
    from torch.utils.data import Sampler
    ns = {'torch': torch, 'random': pyrandom, 'Sampler': Sampler}
    exec('\n\n'.join(chunks), ns)

    class DS:
        def __init__(self, samples):
            self.samples = samples
        def __len__(self):
            return len(self.samples)
        def __getitem__(self, i):
            return self.samples[i]

    out = {'sampler': [], 'collate': []}
    for seed, P, K, reuse in ((0, 3, 4, True), (1, 4, 3, True), (2, 2, 4, False), (3, 5, 2, True)):
        samples = _pipeline_samples(seed)
        sm = ns['ModalAwarePKBatchSampler_Strict'](DS(samples), num_ids_per_batch=P, num_instances=K, allow_id_reuse=reuse)
        pyrandom.seed(100 + seed)
        batches = []
        for b in sm:
            batches.append(list(map(int, b)))
            if len(batches) >= min(6, len(sm)):       # (without id reuse the reference loops forever once its pools run low)
                break
        out['sampler'].append({'seed': seed, 'P': P, 'K': K, 'reuse': reuse, 'rng_seed': 100 + seed, 'len': len(sm),
                               'strong_ids': list(map(int, sm.strong_ids)), 'soft_ids': list(map(int, sm.soft_ids)), 'batches': batches})
        cb = ns['compatible_collate_fn']([samples[i] for i in batches[0]])
        out['collate'].append({'seed': seed, 'indices': batches[0], 'person_id': cb['person_id'].tolist(),
                               'text_description': cb['text_description'], 'modality': cb['modality'],
                               'modality_mask': {m: v.tolist() for m, v in cb['modality_mask'].items()},
                               'image_sums': {m: [float(x) for x in v.double().flatten(1).sum(1)] for m, v in cb['images'].items()},
                               'image_shapes': {m: list(v.shape) for m, v in cb['images'].items()}})
    # Variants (r04): alternate field names, odd captions, odd K, a subset of the indices, no identity reuse with as many batches as
    # are certain to complete -- and what the reference infers per sample.
    out['variants'] = []
    for seed, P, K, reuse in ((10, 3, 4, True), (11, 2, 3, True), (12, 3, 2, False), (13, 2, 5, False), (14, 7, 2, True), (15, 3, 3, True)):
        samples = _pipeline_perturb(_pipeline_samples(seed, n_pid=7), seed)
        keep = [i for i in range(len(samples)) if (i * 7 + seed) % 5 != 0] if seed == 15 else list(range(len(samples)))

        class Sub:                                         # what torch.utils.data.Subset looks like to the sampler
            dataset = DS(samples)
            indices = keep
        sm = ns['ModalAwarePKBatchSampler_Strict'](Sub if seed == 15 else DS(samples), num_ids_per_batch=P, num_instances=K,
                                                   allow_id_reuse=reuse)
        if reuse:
            n = 0 if (len(sm.strong_ids) < P and not sm.soft_ids) else 5
        else:                                              # batches that complete before the reference starts retrying forever
            n = (len(sm.strong_ids) + len(sm.soft_ids)) // P if (not sm.soft_ids or len(sm.strong_ids) % P == 0) else len(sm.strong_ids) // P
        pyrandom.seed(200 + seed)
        import itertools
        batches = [list(map(int, b)) for b in itertools.islice(iter(sm), n)]
        infer = [[sorted(ns['infer_modalities_of_sample'](DS(samples), i, include_text=t)) for t in (True, False)] for i in range(len(samples))]
        case = {'seed': seed, 'P': P, 'K': K, 'reuse': reuse, 'rng_seed': 200 + seed, 'len': len(sm), 'keep': keep,
                'strong_ids': list(map(int, sm.strong_ids)), 'soft_ids': list(map(int, sm.soft_ids)), 'batches': batches, 'infer': infer}
        if batches:
            cb = ns['compatible_collate_fn']([samples[i] for i in batches[0]])
            case['collate'] = {'indices': batches[0], 'person_id': cb['person_id'].tolist(), 'text_description': cb['text_description'],
                               'modality': cb['modality'], 'modality_mask': {m: v.tolist() for m, v in cb['modality_mask'].items()},
                               'image_sums': {m: [float(x) for x in v.double().flatten(1).sum(1)] for m, v in cb['images'].items()}}
        out['variants'].append(case)
    json.dump(out, open(os.path.join(HERE, 'pipeline_cases.json'), 'w'))
    print(f"[pipeline] {sum(len(c['batches']) for c in out['sampler'])} sampler batches, {len(out['collate'])} collate cases, "
          f"{len(out['variants'])} variants with {sum(len(c['batches']) for c in out['variants'])} batches")


# --------------------------------------------------------------------------- checkpoint layout
def make_state_keys():
    """state_keys.json: every key / shape / dtype of the reference model's state_dict() (tiny and full architecture) and, for
    the keys the hot path never reads, which live tensor they are a construction-time copy of (found by comparing values in a
    freshly built reference model, before the seeded overwrite) -- the pin of prcv2025reid_amd/checkpoint.py."""
    import json
    out = {}
    for name, cfg, C in (('tiny', tiny_cfg(4), 5), ('full', full_cfg(8), 16)):
        from transformers import CLIPModel  # noqa: F401  (build_reference rebinds the loader)
        import models.clip_backbone as cb  # noqa: F401
        # same construction as build_reference but WITHOUT the seeded overwrite: copies are still equal
        torch.manual_seed(1234)
        import importlib
        m = build_reference.__wrapped__(cfg, C) if hasattr(build_reference, '__wrapped__') else _build_plain(cfg, C)
        sd = m.state_dict()
        keys = [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in sd.items()]
        from prcv2025reid_amd.weights import is_dead_key
        live = {k: v for k, v in sd.items() if not is_dead_key(k)}
        src = {}
        for k, v in sd.items():
            if not is_dead_key(k):
                continue
            hit = None
            for lk, lv in live.items():
                if lv.numel() == v.numel() and v.numel() > 1 and torch.equal(lv.reshape(-1).float(), v.reshape(-1).float()):
                    hit = lk
                    break
            if hit is None:
                vv = v.float().reshape(-1)
                hit = 'const:%r' % float(vv[0]) if bool((vv == vv[0]).all()) else 'free'
            src[k] = hit
        out[name] = {'keys': keys, 'dead_source': src}
        print(f'[state_keys/{name}] {len(keys)} keys, {len(src)} dead: '
              f'{sum(1 for x in src.values() if not x.startswith(("const", "free")))} copies, '
              f'{sum(1 for x in src.values() if x.startswith("const"))} constants, {sum(1 for x in src.values() if x == "free")} free')
    json.dump(out, open(os.path.join(HERE, 'state_keys.json'), 'w'), indent=0)


def _build_plain(cfg, num_classes):
    """build_reference without seeded_fill (construction-time values as the reference produces them)."""
    import prcv2025reid_amd.weights as W
    orig = W.seeded_fill
    try:
        globals()['seeded_fill'] = lambda sd, seed: None
        globals()['RANDOMIZE_HF'] = True
        return build_reference(cfg, num_classes, 0)
    finally:
        globals()['seeded_fill'] = orig
        globals()['RANDOMIZE_HF'] = False


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None, choices=[None, 'tiny', 'full', 'full16', 'retrieval', 'statekeys', 'pipeline', 'edge', 'groups'])
    args = ap.parse_args()
    torch.set_num_threads(8)
    if args.only in (None, 'retrieval'):
        make_retrieval_cases()
    if args.only in (None, 'tiny', 'full', 'full16'):
        make_model_cases(args.only)
    if args.only in (None, 'edge'):
        make_edge_cases()
    if args.only in (None, 'groups'):
        make_param_groups()
    if args.only in (None, 'statekeys'):
        make_state_keys()
    if args.only in (None, 'pipeline'):
        make_pipeline_cases()
