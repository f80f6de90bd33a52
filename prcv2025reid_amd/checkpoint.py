"""Checkpoint compatibility with the reference.

Reference: ``save_checkpoint`` (train.py:1785-1796) stores ``{'epoch', 'model_state_dict', 'optimizer_state_dict',
'scheduler_state_dict', 'best_map', 'num_classes', 'config'}`` with ``model.state_dict()`` of
``CLIPBasedMultiModalReIDModel``.  That state dict carries 205 tensors the hot path never reads ("dead keys",
SURVEY.md section 9): the whole HF ``clip_model.vision_model`` tower (its weights are copied into the MER blocks at
construction, clip_backbone.py:222-252, and then left behind), ``logit_scale`` / ``visual_projection`` /
``text_projection``, the unused ``channel_adapter`` convolutions of the one-channel patch embeds (patch_embeds.py:38-41)
and BatchNorm's ``num_batches_tracked``.

``full_state_dict`` re-creates those keys from the live tensors by the same rules (pinned by tests/golden/state_keys.json,
generated from the reference itself), so a checkpoint written here loads with ``strict=True`` in the reference, and a
reference checkpoint loads here (dead keys are read, kept for the round trip and otherwise ignored).
``load_clip_pretrained`` does what ``from_pretrained(<hub name>)`` + ``_init_weights`` do, from a LOCAL directory
(``pytorch_model.bin`` or ``model.safetensors``): there is no network in either container.
"""
import math
import os
import re
from collections import OrderedDict
from typing import Dict, Optional

import torch

from .weights import is_dead_key

_VM = 'clip_encoder.clip_model.vision_model.'
_CE = 'clip_encoder.'


def dead_source(key: str) -> Optional[str]:
    """Live key a dead key is a construction-time copy of (None: constant / free)."""
    if key == _VM + 'embeddings.class_embedding':
        return _CE + 'cls_token'
    if key == _VM + 'embeddings.patch_embedding.weight':
        return _CE + 'patch_embeds.vis.proj.weight'
    if key == _VM + 'embeddings.position_embedding.weight':
        return _CE + 'vision_pos_embed'
    m = re.match(re.escape(_VM) + r'encoder\.layers\.(\d+)\.(.+)$', key)
    if m:
        l, rest = m.group(1), m.group(2)
        rest = re.sub(r'^self_attn\.(q|k|v|out)_proj\.(weight|bias)$', r'attn.\1_proj.shared_linear.\2', rest)
        rest = re.sub(r'^mlp\.(fc[12])\.(weight|bias)$', r'mlp.\1.shared_linear.\2', rest)
        rest = re.sub(r'^layer_norm([12])\.(weight|bias)$', r'ln\1.\2', rest)
        return f'{_CE}vision_layers.{l}.{rest}'
    m = re.match(re.escape(_VM) + r'post_layernorm\.(weight|bias)$', key)
    if m:
        return _CE + 'vision_ln_final.' + m.group(1)
    if key == _CE + 'clip_model.visual_projection.weight':
        return _CE + 'vision_proj.weight'
    if key == _CE + 'clip_model.text_projection.weight':
        return _CE + 'text_proj.weight'
    return None


def dead_keys(arch: dict):
    """(key, shape) of every dead tensor of the reference state dict for this architecture."""
    d, ff, L = arch['vision_hidden_dim'], arch['vision_mlp_dim'], arch['vision_layers']
    n_tok = (arch['image_size'] // arch['patch_size']) ** 2 + 1
    out = [(_CE + 'clip_model.logit_scale', ()),
           (_VM + 'embeddings.class_embedding', (d,)),
           (_VM + 'embeddings.patch_embedding.weight', (d, 3, arch['patch_size'], arch['patch_size'])),
           (_VM + 'embeddings.position_embedding.weight', (n_tok, d)),
           (_VM + 'pre_layrnorm.weight', (d,)), (_VM + 'pre_layrnorm.bias', (d,))]
    for l in range(L):
        p = f'{_VM}encoder.layers.{l}.'
        for n in ('k', 'v', 'q', 'out'):
            out += [(p + f'self_attn.{n}_proj.weight', (d, d)), (p + f'self_attn.{n}_proj.bias', (d,))]
        out += [(p + 'layer_norm1.weight', (d,)), (p + 'layer_norm1.bias', (d,)),
                (p + 'mlp.fc1.weight', (ff, d)), (p + 'mlp.fc1.bias', (ff,)),
                (p + 'mlp.fc2.weight', (d, ff)), (p + 'mlp.fc2.bias', (d,)),
                (p + 'layer_norm2.weight', (d,)), (p + 'layer_norm2.bias', (d,))]
    out += [(_VM + 'post_layernorm.weight', (d,)), (_VM + 'post_layernorm.bias', (d,)),
            (_CE + 'clip_model.visual_projection.weight', (arch['fusion_dim'], d)),
            (_CE + 'clip_model.text_projection.weight', (arch['fusion_dim'], arch['text_hidden_dim'])),
            (_CE + 'patch_embeds.nir.channel_adapter.weight', (3, 1, 1, 1)),
            (_CE + 'patch_embeds.sk.channel_adapter.weight', (3, 1, 1, 1)),
            ('bn_neck.bn.num_batches_tracked', ())]
    return out


def full_state_dict(model, kept_dead: Optional[Dict[str, torch.Tensor]] = None) -> 'OrderedDict[str, torch.Tensor]':
    """The reference's ``model.state_dict()``: live keys from the model + the dead keys (the ones kept from a loaded
    reference checkpoint where available, otherwise re-created by the construction rules)."""
    live = model.state_dict()
    kept = kept_dead if kept_dead is not None else getattr(model, '_dead_state', {})
    out = OrderedDict((k, v.detach().cpu().clone()) for k, v in live.items())
    for k, shp in dead_keys(model.arch):
        if k in kept:
            out[k] = kept[k].detach().cpu().clone()
            continue
        src = dead_source(k)
        if src is not None:
            out[k] = out[src].reshape(shp).clone()
        elif k.endswith('logit_scale'):
            out[k] = torch.tensor(math.log(1.0 / 0.07))               # HF CLIP default (logit_scale_init_value 2.6592)
        elif k.endswith('channel_adapter.weight'):
            out[k] = torch.full(shp, 1.0 / 3.0)                       # patch_embeds.py:41
        elif k.endswith('num_batches_tracked'):
            out[k] = torch.tensor(0, dtype=torch.long)
        elif k.endswith('pre_layrnorm.weight'):
            out[k] = torch.ones(shp)                                  # never copied anywhere by the reference: identity LN
        else:
            out[k] = torch.zeros(shp)
    return out


def save_checkpoint(model, optimizer, scheduler_state, epoch: int, best_map: float, config, filename: str):
    """train.py:1785-1796, same dictionary.  ``optimizer`` may be a FusedAdamW (its moments are written in
    torch.optim.AdamW's state-dict layout) or any torch optimizer."""
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    if hasattr(optimizer, 'exp_avg'):        # prcv2025reid_amd.trainer.FusedAdamW
        state, groups, i = {}, [], 0
        for g in optimizer.param_groups:
            idx = []
            for _ in g['params']:
                state[i] = {'step': torch.tensor(float(optimizer.step_count)), 'exp_avg': optimizer.exp_avg[i].detach().cpu(),
                            'exp_avg_sq': optimizer.exp_avg_sq[i].detach().cpu()}
                idx.append(i); i += 1
            groups.append({**{k: v for k, v in g.items() if k != 'params'}, 'params': idx, 'betas': optimizer.betas,
                           'eps': optimizer.eps, 'amsgrad': False})
        opt_state = {'state': state, 'param_groups': groups}
    else:
        opt_state = optimizer.state_dict() if optimizer is not None else None
    ckpt = {'epoch': epoch, 'model_state_dict': full_state_dict(model), 'optimizer_state_dict': opt_state,
            'scheduler_state_dict': scheduler_state, 'best_map': best_map,
            'num_classes': getattr(config, 'num_classes', None) or model.num_classes,
            'config': dict(config.__dict__) if hasattr(config, '__dict__') else str(config)}
    torch.save(ckpt, filename)
    return ckpt


def load_checkpoint(filename: str, model, optimizer=None, strict: bool = True) -> dict:
    """Load a checkpoint in the reference's layout (written by the reference or by ``save_checkpoint``)."""
    ckpt = torch.load(filename, map_location='cpu', weights_only=False)
    sd = ckpt['model_state_dict'] if 'model_state_dict' in ckpt else ckpt
    nc = ckpt.get('num_classes') if isinstance(ckpt, dict) else None
    if nc is None and 'bn_neck.classifier.weight' in sd:
        nc = sd['bn_neck.classifier.weight'].shape[0]
    if nc is not None and model.num_classes != nc:
        model.set_num_classes(int(nc))
    model.load_state_dict(sd, strict=strict)
    model._dead_state = {k: v for k, v in sd.items() if is_dead_key(k)}
    if optimizer is not None and ckpt.get('optimizer_state_dict') and hasattr(optimizer, 'exp_avg'):
        st = ckpt['optimizer_state_dict']['state']
        for i in range(len(optimizer.params)):
            if i in st:
                optimizer.exp_avg[i].copy_(st[i]['exp_avg']); optimizer.exp_avg_sq[i].copy_(st[i]['exp_avg_sq'])
                optimizer.step_count = int(st[i]['step'])
    return {k: v for k, v in ckpt.items() if k not in ('model_state_dict', 'optimizer_state_dict')} if isinstance(ckpt, dict) else {}


# ------------------------------------------------------------------------------------------------------------------
def _read_hf_weights(local_dir: str) -> Dict[str, torch.Tensor]:
    st = os.path.join(local_dir, 'model.safetensors')
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    pt = os.path.join(local_dir, 'pytorch_model.bin')
    if os.path.exists(pt):
        return torch.load(pt, map_location='cpu', weights_only=True)
    raise FileNotFoundError(f'{local_dir}: neither model.safetensors nor pytorch_model.bin (a local HF CLIP directory is needed; '
                            'the hub cannot be reached)')


def clip_to_reference_state(hf: Dict[str, torch.Tensor], arch: dict) -> 'OrderedDict[str, torch.Tensor]':
    """HF ``CLIPModel`` weights -> the reference model's keys, as construction does it (clip_backbone.py:166-252,
    patch_embeds.py:78-105, mer_lora.py load_clip_block_weights): the text tower keeps its names under
    ``clip_encoder.clip_model.``, the vision tower is copied into the MER blocks (and also kept as dead keys), the patch
    embedding goes to all four modalities (three-channel ones verbatim, one-channel ones as the channel mean)."""
    out = OrderedDict()
    for k, v in hf.items():
        if k.endswith('position_ids'):
            continue
        out[_CE + 'clip_model.' + k] = v.clone()
    live_from_dead = {}
    for k, _ in dead_keys(arch):
        src = dead_source(k)
        if src is not None and k in out:
            live_from_dead[src] = out[k]
    for lk, v in live_from_dead.items():
        if lk.endswith('cls_token'):
            v = v.reshape(1, 1, -1)
        out[lk] = v.clone()
    pw = out[_VM + 'embeddings.patch_embedding.weight']
    for m in ('vis', 'cp'):
        out[f'{_CE}patch_embeds.{m}.proj.weight'] = pw.clone()
    for m in ('nir', 'sk'):
        out[f'{_CE}patch_embeds.{m}.proj.weight'] = pw.mean(dim=1, keepdim=True)
    return out


def load_clip_pretrained(model, local_dir: str, noise_seed: Optional[int] = None):
    """Initialise ``model`` from a local HF CLIP directory the way the reference initialises itself from the hub
    (clip_backbone.py:166-252): CLIP tensors are copied in; the non-'vis' patch convolutions additionally get the reference's
    N(0, 0.02^2) weight noise (patch_embeds.py:150-167; seeded here by ``noise_seed``, default config.seed).  Tensors CLIP
    does not provide keep their current values -- for a freshly constructed model (``config.init = 'reference'``) those are the
    reference's own construction values: lora_B = 0 / lora_A kaiming-uniform (initial low-rank update exactly 0), xavier SDM
    module, default fusion block, BN-neck (1, 0), null tokens N(0, 0.02^2)."""
    ref = clip_to_reference_state(_read_hf_weights(local_dir), model.arch)
    seed = int(getattr(model.config, 'seed', 42) if noise_seed is None else noise_seed)
    for m in ('nir', 'sk', 'cp'):
        k = f'{_CE}patch_embeds.{m}.proj.weight'
        g = torch.Generator().manual_seed(seed * 1000003 + sum(map(ord, m)))
        ref[k] = ref[k] + torch.randn(ref[k].shape, generator=g) * 0.02
    res = model.load_state_dict(ref, strict=False)
    model._dead_state = {k: v for k, v in ref.items() if is_dead_key(k)}
    return res
