"""Drop-in for the reference's ``models/model.py::CLIPBasedMultiModalReIDModel``.

Same constructor (``config`` object read with getattr), same ``forward(images, texts, modality_masks,
return_features)`` / ``compute_loss(outputs, labels)`` / ``set_num_classes`` / ``set_epoch`` /
``get_learnable_params`` surface, same output-dict keys and the same ``state_dict`` key names for every
tensor the hot path uses (reference: models/model.py:227-737).  Underneath, the
encoders run on the HIP executor in engine.py and the BN-neck / CE / SDM kernels in head.py.  There is
no CPU path: constructing the model needs a gfx950 device and libreid_hip.so.

Deliberate differences (documented in DESIGN.md):
  * weights are random-initialised from ``config.seed`` (prcv2025reid_amd.weights.seeded_state) unless a
    state dict is loaded -- the reference downloads CLIP by name (clip_backbone.py:170), impossible offline;
  * ``texts`` may be ``List[str]`` (tokenised on the host by a local tokenizer) or a dict with
    ``input_ids``/``attention_mask`` (pre-tokenised, keeps the tokenizer off the hot path);
  * all LoRA adapters live in ONE flat fp32 parameter (``clip_encoder.vision_layers.loras.arena``);
    ``state_dict()`` still emits the reference's per-adapter keys, and ``load_state_dict`` accepts them;
  * the stochastic regularisers of the training forward ARE applied in train mode -- DropPath (clip_backbone.py:126-142),
    the dropouts of the SDM module / fusion block / BN-neck (models/model.py:35,43,95,104,106,221) and the batch-level
    modality dropout (models/model.py:434-474) -- from this package's own generators (the random streams cannot coincide
    with torch's: the reference draws per modality pass, here the modalities run packed); modality dropout masks the dropped
    modality's fusion slot instead of removing it from the list (same function, static shapes); the reference's train/eval
    asymmetry (SDM module only in train mode, models/model.py:395-399) is kept;
  * gradients exist for EVERY parameter (freeze_backbone=False works): the LoRA / bn_neck / null-token default set on the
    fast path, and -- only when such a tensor has requires_grad -- the vision backbone (VisionEncodeFn's extra inputs) and the
    text tower (TextEncodeFn); the frozen default carries none of that state or work.
"""
import logging
from collections import OrderedDict
from typing import Any, Dict, List, Optional

import os

import torch
import torch.nn as nn

from . import _lib, ops
from .config import arch_of
from .engine import TextEncodeFn, Engine, LoraLayout, VisionEncodeFn
from .head import (MulFn, ActFn, AddFn, BNNeckFn, CrossEntropyLSFn, LayerNormF32Fn, LinearF32Fn, LinearNdF32Fn, MaskedMeanFn,
                   NanToNumFn, SDMFn, SmallAttnFn)
from .tokenizer import load_tokenizer
from .weights import param_spec, reference_init_state, seeded_tensor, is_dead_key

logger = logging.getLogger(__name__)

LORA_PARAM_NAME = 'clip_encoder.vision_layers.loras.arena'
_REF_LIN = ('attn.q_proj', 'attn.k_proj', 'attn.v_proj', 'attn.out_proj', 'mlp.fc1', 'mlp.fc2')


class LazyCount:
    """Integer that lives on the device until someone looks at it (keeps compute_loss free of host syncs)."""

    def __init__(self, t):
        self._t = t

    def __int__(self):
        return int(self._t.item())

    __index__ = __int__

    def __eq__(self, o):
        return int(self) == o

    def __gt__(self, o):
        return int(self) > o

    def __repr__(self):
        return str(int(self))


class _NS:
    pass


class CLIPBasedMultiModalReIDModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.device = getattr(config, 'device', 'cuda')
        if not torch.cuda.is_available() or not str(self.device).startswith('cuda'):
            raise _lib.ReidHipError('CLIPBasedMultiModalReIDModel needs an MI355X (device="cuda"): the hot path has no CPU fallback')
        self.compute_dtype = getattr(config, 'compute_dtype', None) or _lib.flavor()
        _lib.set_flavor(self.compute_dtype)
        _lib.check(_lib.lib().reid_check_device(torch.cuda.current_device()))
        self.current_epoch = 0
        self.sdm_memory = None
        self.arch = arch_of(config)
        self.modalities = self.arch['modalities']
        self.vision_modalities = [m for m in self.modalities if m != 'text']
        self.fusion_dim = self.arch['fusion_dim']
        self.vision_hidden_dim = self.arch['vision_hidden_dim']
        self.sdm_temperature = getattr(config, 'sdm_temperature', 0.2)
        self.ce_weight = getattr(config, 'ce_weight', 1.0)
        self.contrastive_weight = getattr(config, 'contrastive_weight', 0.1)
        self.num_classes = None
        self.bn_neck = None
        self._ref: "OrderedDict[str, nn.Parameter]" = OrderedDict()
        self._bufs: Dict[str, torch.Tensor] = {}
        seed = getattr(config, 'seed', 42)
        self._seed = seed
        self.layout = LoraLayout(self.arch)
        dev = torch.device(self.device)
        lora_host = torch.zeros(self.layout.size)
        # Initial values: the reference's CONSTRUCTION semantics (weights.reference_init_state: lora_B = 0, lora_A
        # kaiming-uniform, xavier SDM module, CLIP-derived patch convolutions + noise, ...) with seeded stand-ins where the
        # reference would download CLIP; ``config.init = 'seeded'`` selects the all-random seeded fill the parity fixtures
        # use (every tensor non-trivial, lora_B included, so no path is vacuously correct).
        self._init_mode = getattr(config, 'init', 'reference')
        init_state = reference_init_state(self.arch, None, seed) if self._init_mode == 'reference' else None
        for k, shp in param_spec(self.arch, None).items():
            v = init_state[k] if init_state is not None else seeded_tensor(k, shp, seed)
            if '.loras.' in k:
                self._lora_put(lora_host, k, v)
            else:
                self._add_param(k, v.to(dev))
        self.lora_arena = nn.Parameter(lora_host.to(dev))
        self.register_parameter('p/' + LORA_PARAM_NAME.replace('.', '/'), self.lora_arena)
        self._ref[LORA_PARAM_NAME] = self.lora_arena
        self.engine = Engine(self.arch, self._ref, self.lora_arena, dev)
        self._plans = {}
        self._plan_ids = {}
        self._fusion_cache = {}
        self._loss_cache = {}
        self._forced_keep = None       # tests only: fixed keep decisions of the next modality-dropout draws
        self._overlap_text = os.environ.get('REID_TEXT_STREAM', '1') != '0'
        # Stochastic regularisers of the reference's training forward (all inactive in eval mode):
        #   DropPath in the vision blocks (clip_backbone.py:137-141,204), dropout in the SDM module (hard-coded 0.1: model.py:35,43),
        #   in the fusion block (fusion_dropout: model.py:95,104,106), before the classifier (dropout_rate: model.py:200,221),
        #   and batch-level modality dropout (model.py:434-474).
        # Head-level masks come from a generator seeded identically on every data-parallel rank (the head is evaluated
        # redundantly on the gathered global batch, so its random masks must agree); DropPath uses the engine's own generator.
        self.drop_path = float(getattr(config, 'drop_path', 0.0))
        self.dropout_rate = float(getattr(config, 'dropout_rate', 0.5))
        self.fusion_dropout = float(getattr(config, 'fusion_dropout', 0.1))
        self.sdm_dropout = float(getattr(config, 'sdm_dropout', 0.1))
        self._rng_head = torch.Generator(device=dev); self._rng_head.manual_seed(int(seed) + 12345)
        self._rng_host = torch.Generator(); self._rng_host.manual_seed(int(seed) + 54321)
        self.engine._rng.manual_seed(int(seed) + 777)
        self.tokenizer = load_tokenizer(getattr(config, 'clip_model_name', ''), self.arch['text_vocab'],
                                        self.arch['text_bos_id'], self.arch['text_eos_id'], self.arch['text_max_len'])
        # facades so callers written against the reference's attribute paths keep working
        self.clip_encoder = _NS()
        self.clip_encoder.encode_vision = self.encode_vision
        self.clip_encoder.encode_text = self.encode_text
        self.clip_encoder.tokenizer = self.tokenizer
        self.feature_fusion = self._fusion
        self.sdm_module = self._sdm_module

    # ------------------------------------------------------------------ parameter plumbing
    def _add_param(self, ref_name: str, value: torch.Tensor, buffer: bool = False):
        if buffer:
            self._bufs[ref_name] = value
            self.register_buffer('b/' + ref_name.replace('.', '/'), value)
            return
        p = nn.Parameter(value)
        self._ref[ref_name] = p
        self.register_parameter('p/' + ref_name.replace('.', '/'), p)

    def _lora_slot(self, key: str):
        # clip_encoder.vision_layers.{l}.{lin}.loras.{m}.lora_{A|B}.weight
        parts = key.split('.')
        l = int(parts[2]); lin = parts[3] + '.' + parts[4]; m = parts[6]; which = parts[7]
        e, g, mu, n_out = self.layout.ref_slices(l, lin, m)
        r, Rp = self.layout.r, self.layout.Rp
        return e, g, mu, n_out, r, Rp, which

    def _lora_put(self, arena: torch.Tensor, key: str, v: torch.Tensor):
        e, g, mu, n_out, r, Rp, which = self._lora_slot(key)
        if which == 'lora_A':
            o, (rows, K) = e['A']
            arena[o:o + rows * K].view(rows, K)[g * Rp + mu * r: g * Rp + (mu + 1) * r].copy_(v)
        else:
            o, (N, _) = e['B']
            arena[o:o + N * Rp].view(N, Rp)[g * n_out:(g + 1) * n_out, mu * r:(mu + 1) * r].copy_(v)

    def _lora_get(self, arena: torch.Tensor, key: str) -> torch.Tensor:
        e, g, mu, n_out, r, Rp, which = self._lora_slot(key)
        if which == 'lora_A':
            o, (rows, K) = e['A']
            return arena[o:o + rows * K].view(rows, K)[g * Rp + mu * r: g * Rp + (mu + 1) * r]
        o, (N, _) = e['B']
        return arena[o:o + N * Rp].view(N, Rp)[g * n_out:(g + 1) * n_out, mu * r:(mu + 1) * r]

    def lora_grad_view(self, key: str) -> Optional[torch.Tensor]:
        """Gradient of one reference adapter tensor (view into the arena gradient)."""
        return None if self.lora_arena.grad is None else self._lora_get(self.lora_arena.grad, key)

    def named_parameters(self, prefix: str = '', recurse: bool = True, remove_duplicate: bool = True):
        for k, p in self._ref.items():
            yield (prefix + ('.' if prefix else '') + k, p)

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        out = OrderedDict() if destination is None else destination
        arena = self.lora_arena if keep_vars else self.lora_arena.detach()
        for k in param_spec(self.arch, self.num_classes).keys():
            if '.loras.' in k:
                out[prefix + k] = self._lora_get(arena, k).clone()
            elif k in self._ref:
                out[prefix + k] = self._ref[k] if keep_vars else self._ref[k].detach()
            elif k in self._bufs:
                out[prefix + k] = self._bufs[k]
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        want = set(param_spec(self.arch, self.num_classes).keys())
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in want and not is_dead_key(k)]
        if strict and (missing or unexpected):
            raise RuntimeError(f'load_state_dict: missing {missing[:5]}... unexpected {unexpected[:5]}...')
        with torch.no_grad():
            for k, v in state_dict.items():
                if k not in want:
                    continue
                v = torch.as_tensor(v)
                if '.loras.' in k:
                    self._lora_get(self.lora_arena, k).copy_(v.to(self.lora_arena.device))
                elif k in self._ref:
                    self._ref[k].copy_(v.to(self._ref[k].device))
                else:
                    self._bufs[k].copy_(v.to(self._bufs[k].device))
            self.lora_arena.add_(0)          # bump versions so the bf16 packs are rebuilt
            for p in self._ref.values():
                p.add_(0)
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def set_num_classes(self, num_classes: int):
        """models/model.py:310-319: create the BN-neck for ``num_classes`` identities."""
        self.num_classes = num_classes
        dev = torch.device(self.device)
        for k, shp in param_spec(self.arch, num_classes).items():
            if not k.startswith('bn_neck.'):
                continue
            v = seeded_tensor(k, shp, self._seed).to(dev)
            # reference initial values: BN affine (1, 0), running (0, 1), classifier N(0, 0.001^2)
            if k.endswith('bn.weight') or k.endswith('running_var'):
                v = torch.ones_like(v)
            elif k.endswith('bn.bias') or k.endswith('running_mean'):
                v = torch.zeros_like(v)
            elif k.endswith('classifier.weight'):
                v = v * (0.001 / 0.02)
            self._add_param(k, v, buffer='running_' in k)
        self._ref['bn_neck.bn.bias'].requires_grad_(False)        # models/model.py:197
        ns = _NS(); ns.bn = _NS(); ns.classifier = _NS()
        ns.bn.weight = self._ref['bn_neck.bn.weight']; ns.bn.bias = self._ref['bn_neck.bn.bias']
        ns.bn.running_mean = self._bufs['bn_neck.bn.running_mean']; ns.bn.running_var = self._bufs['bn_neck.bn.running_var']
        ns.classifier.weight = self._ref['bn_neck.classifier.weight']
        ns.dropout = nn.Identity()
        self.bn_neck = ns
        logger.info('classifier for %d identities', num_classes)

    def set_epoch(self, epoch: int):
        self.current_epoch = epoch

    def _check_trainable(self):
        for k, p in self._ref.items():
            if p.requires_grad and k.startswith('clip_encoder.clip_model.text_model.') and not self.engine.text_backward_ready:
                raise NotImplementedError(f'gradient of {k} requested: the text tower has no backward pass in this build')

    def _text_apply(self, ids, am):
        """Text tower + projection; through TextEncodeFn (activations saved, backward available) only when one of its tensors
        trains (freeze_backbone=False and freeze_text_backbone=False) -- the reference default runs it forward-only."""
        if torch.is_grad_enabled() and self.engine.text_trains():
            return TextEncodeFn.apply(self.engine, ids, am, *[self._ref[k] for k in self.engine.text_keys()])
        return self.engine.text_forward(ids, am)

    def _vision_apply(self, mods, images):
        """VisionEncodeFn with the backbone tensors as extra autograd inputs only when one of them trains
        (freeze_backbone=False); the reference default (train.py:1418-1425) passes none."""
        if not torch.is_grad_enabled():
            # No graph will be built: run the executor forward-only.  (Function.apply fills ctx.needs_input_grad from the inputs'
            # requires_grad flags whatever the grad mode, so going through VisionEncodeFn here would save every activation and
            # queue the adapter-gradient side products on the side stream -- whose outputs die with the discarded ctx while
            # those kernels are still pending: their blocks were handed to the head's tensors and overwritten under them.)
            return self.engine.vision_forward(list(zip(tuple(mods), images)), save=False)[0]
        dense = []
        keys = self.engine.vision_dense_keys()
        if any(self._ref[k].requires_grad for k in keys):
            dense = [self._ref[k] for k in keys]
        return VisionEncodeFn.apply(self.engine, tuple(mods), self.lora_arena, len(images), *images, *dense)

    # ------------------------------------------------------------------ encoders
    def _tokens(self, texts):
        if isinstance(texts, dict):
            return texts['input_ids'], texts.get('attention_mask')
        t = self.tokenizer(list(texts), return_tensors='pt', padding=True, truncation=True, max_length=self.arch['text_max_len'])
        return t['input_ids'], t['attention_mask']

    def encode_vision(self, images: torch.Tensor, modality: str) -> torch.Tensor:
        """clip_backbone.py:254-286 for one modality."""
        _lib.set_flavor(self.compute_dtype)
        self.engine.refresh()
        mu = self.vision_modalities.index(modality)
        return self._vision_apply((mu,), [images.to(self.device).float()])

    def encode_text(self, texts) -> torch.Tensor:
        """clip_backbone.py:288-313."""
        _lib.set_flavor(self.compute_dtype)
        self.engine.refresh()
        ids, am = self._tokens(texts)
        out = self._text_apply(ids, am)
        self.engine.wait_packed()
        return out

    def seed_stochastic(self, seed: int, rank: int = 0):
        """Reseed the regularisers' generators (head masks identical on every rank, DropPath per rank)."""
        self._rng_head.manual_seed(int(seed) + 12345)
        self._rng_host.manual_seed(int(seed) + 54321)
        self.engine._rng.manual_seed(int(seed) + 777 + 1000003 * int(rank))

    def _keep_mask(self, shape, p: float) -> torch.Tensor:
        """Dropout multipliers: 0 with probability p, else 1 / (1 - p)."""
        u = torch.rand(shape, device=self._rng_head.device, generator=self._rng_head)
        return ops.eltwise('keep_mask', u, out=u, alpha=float(p))            # in place: one launch instead of compare, cast, divide

    def _dropout(self, x, p: float):
        if not self.training or p <= 0.0:
            return x
        return MulFn.apply(x, self._keep_mask(tuple(x.shape), p))

    # ------------------------------------------------------------------ head modules ([B,512] / [B,M,512], fp32 HIP kernels)
    def _sdm_module(self, x):
        """SemanticDisentanglementModule.forward, models/model.py:57-77 (length-1 MHA == out_proj(v_proj(x)))."""
        P, D = self._ref, self.fusion_dim
        lin = LinearNdF32Fn.apply
        wv = P['sdm_module.semantic_attn.in_proj_weight'][2 * D:]; bv = P['sdm_module.semantic_attn.in_proj_bias'][2 * D:]
        v = lin(x, wv, bv)
        if self.training and self.sdm_dropout > 0:
            # nn.MultiheadAttention(dropout=0.1) on a length-1 sequence: the single attention weight (= 1) of every
            # (sample, head) is dropped or scaled by 1/keep, i.e. that head's 64 value channels are
            H = self.arch['sdm_num_heads']
            m = self._keep_mask((x.shape[0], H, 1), self.sdm_dropout).expand(x.shape[0], H, D // H).reshape(x.shape[0], D)
            v = MulFn.apply(v, m)
        a = lin(v, P['sdm_module.semantic_attn.out_proj.weight'], P['sdm_module.semantic_attn.out_proj.bias'])
        y = lin(AddFn.apply(x, a), P['sdm_module.semantic_proj.0.weight'], P['sdm_module.semantic_proj.0.bias'])
        y = LayerNormF32Fn.apply(y, P['sdm_module.semantic_proj.1.weight'], P['sdm_module.semantic_proj.1.bias'], 1e-5)
        y = self._dropout(ActFn.apply(y, 'relu'), self.sdm_dropout)
        return lin(y, P['sdm_module.semantic_proj.4.weight'], P['sdm_module.semantic_proj.4.bias'])

    def _fusion(self, features: List[torch.Tensor], masks: Optional[List[torch.Tensor]] = None, stable_masks: bool = False):
        """FeatureFusion.forward, models/model.py:113-183.  stack / where / cat below only move data; every arithmetic
        step is a HIP kernel (fp32)."""
        if len(features) == 0:
            raise ValueError('No features to fuse')
        if len(features) == 1:
            return features[0]
        P, D = self._ref, self.fusion_dim
        heads = self.arch['fusion_num_heads']
        lin, ln = LinearNdF32Fn.apply, LayerNormF32Fn.apply
        x = torch.stack(features, dim=1).float()
        B, M, _ = x.shape
        sm = None; km = None
        if masks is not None:
            # Everything below that depends on the masks alone is derived ONCE per mask set (the routing plan hands out the same device
            # tensors for a repeated pattern): the stacked mask, the key-padding mask with the all-masked-row rule of model.py:141-149
            # applied, and whether any sample has no valid modality at all (a host flag read once per new pattern).  The head section
            # is host-bound (~100 launches of 2-5 us, 1.9 ms end to end in the r03 trace): every launch not issued there is ~15 us.
            key = tuple(id(t) for t in masks)
            ent = self._fusion_cache.get(key) if stable_masks else None
            if ent is None or any(a is not b for a, b in zip(ent['masks'], masks)):
                smc = torch.stack(masks, dim=1).to(x.device).float()
                pad = smc <= 0
                dead = pad.all(dim=1)
                padk = torch.cat([(pad[:, 0] & ~dead).unsqueeze(1), pad[:, 1:]], dim=1)
                ent = dict(masks=list(masks), sm=smc, km=(~padk).to(torch.uint8).contiguous(), dead=dead,
                           live=(~dead).float().view(1, B, 1).expand(1, B, M).reshape(1, B * M).contiguous(),
                           # masks that change from step to step (modality dropout, the data-parallel gather) are never read back:
                           # the all-masked-row path then always runs (it is exact either way), as before
                           has_dead=bool(dead.any()) if stable_masks else True)
                if stable_masks:
                    if len(self._fusion_cache) > 16:
                        self._fusion_cache.clear()
                    self._fusion_cache[key] = ent
            sm, km = ent['sm'], ent['km']
            if ent['has_dead']:
                # all-masked rows: unmask slot 0 and put the mean of the live rows there (model.py:141-149):
                # gm = masked mean over all (sample, slot) pairs of live samples
                gm = MaskedMeanFn.apply(x.reshape(1, B * M, D), ent['live'])              # [1, D]
                x = torch.cat([torch.where(ent['dead'].view(B, 1), gm.expand(B, D), x[:, 0]).unsqueeze(1), x[:, 1:]], dim=1)
        qkv = lin(x.reshape(B * M, D), P['feature_fusion.multihead_attn.in_proj_weight'], P['feature_fusion.multihead_attn.in_proj_bias'])
        drop = None
        if self.training and self.fusion_dropout > 0:
            drop = self._keep_mask((B, heads, 8, 8), self.fusion_dropout)            # attention-probability dropout (model.py:95)
        a = SmallAttnFn.apply(qkv, km, B, M, heads, drop)
        a = lin(a, P['feature_fusion.multihead_attn.out_proj.weight'], P['feature_fusion.multihead_attn.out_proj.bias'])
        y = ln(AddFn.apply(x.reshape(B * M, D), a), P['feature_fusion.norm1.weight'], P['feature_fusion.norm1.bias'], 1e-5)
        m = ln(y, P['feature_fusion.mlp.0.weight'], P['feature_fusion.mlp.0.bias'], 1e-5)
        m = self._dropout(ActFn.apply(lin(m, P['feature_fusion.mlp.1.weight'], P['feature_fusion.mlp.1.bias']), 'gelu'), self.fusion_dropout)
        m = self._dropout(lin(m, P['feature_fusion.mlp.4.weight'], P['feature_fusion.mlp.4.bias']), self.fusion_dropout)
        z = ln(AddFn.apply(y, m), P['feature_fusion.norm2.weight'], P['feature_fusion.norm2.bias'], 1e-5)
        z = NanToNumFn.apply(z).view(B, M, D)
        if sm is None:
            sm = torch.ones(B, M, device=x.device)
        return MaskedMeanFn.apply(z, sm)

    def _modality_dropout(self, names: List[str], masks: List[torch.Tensor]):
        """Batch-level modality dropout, models/model.py:434-474 -> (masks after the draw, keep flags or None, device flag).

        The reference draws ``torch.rand(1).item() > p`` once per non-'vis' modality (in order), removes the dropped ones from
        the fused list AND from ``feature_masks`` (so compute_loss skips their SDM pairs and counts validity over the kept
        ones), cancels the whole draw if a sample would be left without any valid modality, and returns the single survivor
        unfused when only 'vis' is kept.  Here the same function runs with static shapes and no host read-back: a dropped
        modality's mask becomes all-zero (a masked slot is no attention key, its own output is excluded by the masked mean, an
        all-zero mask makes compute_loss skip the modality exactly like a missing key), and the cancel rule is a device flag
        ``ok`` that selects between the dropped and the original masks.  Inactive while epoch <= warm-up epochs."""
        cfg = self.config
        p = float(getattr(cfg, 'modality_dropout', 0.0))
        min_mod = int(getattr(cfg, 'min_modalities', 1))
        if self.current_epoch <= int(getattr(cfg, 'modality_dropout_warmup_epochs', 3)):
            p = 0.0
        if p <= 0.0 or len(names) <= min_mod:
            return masks, None, None
        keep = self._forced_keep if self._forced_keep is not None else \
            [m == 'vis' or float(torch.rand(1, generator=self._rng_host)) > p for m in names]
        keep = [bool(k) or m == 'vis' for k, m in zip(keep, names)]
        if all(keep) or sum(keep) < min_mod:
            return masks, None, None
        dev = masks[0].device
        kv = self.engine._const(('moddrop', tuple(keep)), lambda: torch.tensor([1.0 if k else 0.0 for k in keep]))
        stacked = torch.stack([m.to(dev).float() for m in masks], dim=1)             # [B, M]
        dropped = stacked * kv.view(1, -1)
        ok = ((dropped > 0).any(dim=1)).all()                                        # every sample keeps >= 1 valid modality
        final = torch.where(ok, dropped, stacked)
        return [final[:, i].contiguous() for i in range(len(masks))], keep, ok

    def _plan(self, images, modality_masks, B):
        """{modality: (kind 'all' | 'some' | 'none', device row indices or None, device mask f32 [B])}.

        Which rows of which modality go through the encoder sizes the packed batch, so it has to be known on the HOST.  Masks
        given as host tensors (what the reference's collate produces) cost nothing.  Device-resident masks (train.py:742 moves
        the whole batch) are read back with ONE stacked device->host copy for all modalities (the reference itself synchronises
        five times here: ``mask.sum() > 0`` per modality, model.py:367); with ``trust_mask_identity`` (opt-in, see below) they
        are first looked up by tensor identity + version."""
        dev = torch.device(self.device)
        names = [m for m in (images or {}) if m in self.vision_modalities] + ['text']
        given = {m: (None if modality_masks is None else modality_masks.get(m)) for m in names}
        ident = (B, tuple((m, None if t is None else ((t.data_ptr(), t._version, tuple(t.shape), str(t.device)) if torch.is_tensor(t)
                                                     else id(t))) for m, t in given.items()))
        on_dev = [m for m, t in given.items() if torch.is_tensor(t) and t.is_cuda]
        # The identity cache is OPT-IN (``model.trust_mask_identity = True``): (data_ptr, _version, shape) does not change when a
        # mask buffer is rewritten behind autograd's back (a custom kernel, a graph replay filling a static buffer,
        # ``mask.data.copy_()``), and a stale plan would silently route the wrong rows through the encoders.  Callers that own
        # their mask tensors and modify them only through version-counted in-place ops may switch it on to skip the read-back.
        if on_dev and getattr(self, 'trust_mask_identity', False):
            hit = self._plan_ids.get(ident)
            if hit is not None and all(given[m] is hit[1][m] for m in on_dev):       # same tensor objects, unmodified
                return hit[0]
        host = {}
        if on_dev:
            stacked = torch.stack([given[m].detach().float().reshape(-1) for m in on_dev], dim=0).to('cpu')   # one D2H copy
            for i, m in enumerate(on_dev):
                host[m] = stacked[i].contiguous()
        for m, t in given.items():
            if m in host:
                continue
            host[m] = None if t is None else (t.detach() if torch.is_tensor(t) else torch.as_tensor(t)).float().contiguous()
        key = (B, tuple((m, None if h is None else h.numpy().tobytes()) for m, h in host.items()))
        plan = self._plans.get(key)
        if plan is None:
            plan = {}
            for m, h in host.items():
                if h is None:        # no mask given: text counts as present, an image modality without a mask is absent (model.py:367)
                    plan[m] = ('all', None, torch.ones(B, device=dev)) if m == 'text' else ('none', None, torch.zeros(B, device=dev))
                elif float(h.sum()) <= 0 and m != 'text':
                    plan[m] = ('none', None, torch.zeros(B, device=dev))
                elif bool(h.bool().all()):
                    plan[m] = ('all', None, h.to(dev))
                else:
                    plan[m] = ('some', h.bool().nonzero().flatten().to(dev), h.to(dev))
            if len(self._plans) > 32:
                self._plans.clear()
            self._plans[key] = plan
        if on_dev and getattr(self, 'trust_mask_identity', False):
            if len(self._plan_ids) > 8:
                self._plan_ids.clear()
            self._plan_ids[ident] = (plan, {m: given[m] for m in on_dev})               # keeps the tensors alive: ids stay unique
        return plan

    # ------------------------------------------------------------------ forward
    def forward(self, images: Optional[Dict[str, torch.Tensor]] = None, texts=None,
                modality_masks: Optional[Dict[str, torch.Tensor]] = None, return_features: bool = False,
                gather_fn=None) -> Dict[str, Any]:
        """models/model.py:321-510.  ``gather_fn`` (data parallel only) maps (raw features, masks) of this rank's
        rows to the global batch before the head (prcv2025reid_amd.parallel)."""
        dev = torch.device(self.device)
        batch_size = None
        if images is not None:
            for t in images.values():
                batch_size = t.shape[0]
                break
        elif texts is not None:
            batch_size = texts['input_ids'].shape[0] if isinstance(texts, dict) else len(texts)
        if batch_size is None:
            raise ValueError('cannot determine batch size')
        if torch.is_grad_enabled():
            self._check_trainable()
        _lib.set_flavor(self.compute_dtype)
        self.engine.refresh()
        B = batch_size
        # Routing plan: which rows of which modality go through the encoder is decided from ONE host copy of the masks (no
        # per-modality .sum() syncs as in model.py:367) and cached per mask pattern, together with the device copies of the
        # index / mask tensors -- a repeated pattern (all-on masks, or a replayed HIP graph) costs no host<->device traffic.
        plan = self._plan(images, modality_masks, B)
        raw: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        fmask: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        groups, order = [], []
        if images is not None:
            for m, img in images.items():
                if m not in self.vision_modalities:
                    continue
                kind, sel_idx, mask_dev = plan[m]
                if kind == 'none':
                    order.append((m, 'none', mask_dev))
                else:
                    sel = img.to(dev) if kind == 'all' else img.to(dev)[sel_idx]
                    groups.append((self.vision_modalities.index(m), sel.float()))
                    order.append((m, None if kind == 'all' else sel_idx, mask_dev))
        # The (frozen) text tower is independent of the vision pass and made of small launches (B*T ~ 5k rows): it runs on a
        # second HIP stream underneath the vision encoder's big GEMMs and is joined before the head.
        n_text = 0 if texts is None else (texts['input_ids'].shape[0] if isinstance(texts, dict) else len(texts))
        tf = None; text_ev = None
        if texts is not None and n_text > 0:
            ids, am = self._tokens(texts)
            if self._overlap_text and not (torch.is_grad_enabled() and self.engine.text_trains()) and groups:
                main = torch.cuda.current_stream(dev)
                side = self.engine._text_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    tf = self.engine.text_forward(ids.to(dev), None if am is None else am.to(dev))
                text_ev = torch.cuda.Event(); text_ev.record(side)
            else:
                tf = self._text_apply(ids, am)
        feats = None
        if groups:
            if self.training and self.drop_path > 0 and torch.is_grad_enabled():
                self.engine.pending_drop_scales = self.engine.drop_path_scales(sum(g[1].shape[0] for g in groups), self.drop_path)
            feats = self._vision_apply(tuple(g[0] for g in groups), [g[1] for g in groups])
        start = 0
        for m, idx, mask in order:
            null = self._ref[f'null_tokens.{m}']
            if isinstance(idx, str):
                full = null.expand(B, -1)
            elif idx is None:
                full = feats[start:start + B]; start += B
            else:
                n = idx.shape[0]
                full = null.expand(B, -1).clone().index_copy(0, idx, feats[start:start + n]); start += n
            raw[m] = full; fmask[m] = mask
        if tf is not None:
            if text_ev is not None:
                torch.cuda.current_stream(dev).wait_event(text_ev)
                tf.record_stream(torch.cuda.current_stream(dev))
            tmd = plan['text'][2]
            if plan['text'][0] != 'all':
                tf = torch.where(tmd.bool().view(B, 1), tf, self._ref['null_tokens.text'].expand(B, -1))
            raw['text'] = tf; fmask['text'] = tmd
        if not raw:
            raise ValueError('at least one modality is required')
        # a step without any vision pass (text-only batch) has not joined the weight pack of engine.refresh(): the optimizer must not
        # write the adapter arena while the pack stream still reads it (no-op when the vision pass has already waited)
        self.engine.wait_packed()
        if gather_fn is not None:
            raw, fmask = gather_fn(raw, fmask)
        plan_ids = {id(v[2]) for v in plan.values()} if gather_fn is None else set()
        return self.head_section(raw, fmask, return_features=return_features, plan_mask_ids=plan_ids)

    def head_section(self, raw, fmask, return_features: bool = False, plan_mask_ids=frozenset()):
        """Everything of forward() behind the encoders (model.py:392-510): SDM module per modality (training), batch-level modality
        dropout, fusion, BN-neck, classifier -- on per-modality [B, D] features with the null tokens already filled in.  Under data
        parallelism B is the GLOBAL batch (parallel.py gathers the features first), which every rank evaluates redundantly; bench.py
        times this section alone at that size.  ``plan_mask_ids``: ids of mask tensors owned by the cached routing plan."""
        if self.training:
            # the SDM module is shared by all modalities (model.py:395-399): one pass over the stacked [n_mod*B, D] rows instead
            # of one per modality (same arithmetic per row, 5x fewer of the ~20 tiny launches of its forward + backward)
            names = list(raw.keys())
            Bg = raw[names[0]].shape[0]
            ys = self._sdm_module(torch.cat([raw[m] for m in names], dim=0))
            sem = OrderedDict((m, ys[i * Bg:(i + 1) * Bg]) for i, m in enumerate(names))
        else:
            sem = OrderedDict(raw.items())
        flist = list(sem.values()); mlist = [fmask[m] for m in sem]
        keep = ok = None
        if self.training:
            mlist, keep, ok = self._modality_dropout(list(sem.keys()), mlist)
        stable = keep is None and bool(plan_mask_ids) and all(id(t) in plan_mask_ids for t in mlist)      # mask tensors owned by the cached routing plan
        fused = flist[0] if len(flist) == 1 else self._fusion(flist, mlist, stable_masks=stable)
        if keep is not None:
            # masks after the draw are what compute_loss must see (model.py:466-468: dropped modalities leave feature_masks)
            fmask = OrderedDict((m, mlist[i]) for i, m in enumerate(sem.keys()))
            if sum(keep) == 1:        # only 'vis' survives: the reference returns it unfused (model.py:479-480) unless the draw is cancelled
                fused = torch.where(ok, flist[list(sem.keys()).index('vis')], fused)
        out = {'features': fused, 'raw_modality_features': raw, 'modality_features': sem}
        if self.bn_neck is not None:
            P = self._ref
            bnf = BNNeckFn.apply(fused, P['bn_neck.bn.weight'], P['bn_neck.bn.bias'], self._bufs['bn_neck.bn.running_mean'],
                                 self._bufs['bn_neck.bn.running_var'], self.training, 0.1, 1e-5)
            out['bn_features'] = bnf
            out['logits'] = LinearF32Fn.apply(self._dropout(bnf, self.dropout_rate), P['bn_neck.classifier.weight'], None)
        if return_features:
            out['intermediate_features'] = {'raw_modality': raw, 'semantic_modality': sem, 'fused': fused}
        out['feature_masks'] = fmask
        return out

    # ------------------------------------------------------------------ loss
    def compute_loss(self, outputs: Dict[str, Any], labels: torch.Tensor) -> Dict[str, Any]:
        """models/model.py:512-659 without host syncs: validity masks and 'no positives' cases are device flags."""
        if 'logits' not in outputs:
            raise ValueError('outputs lack logits: call set_num_classes first')
        logits = outputs['logits']
        dev = logits.device
        labels = labels.to(dev).long()
        fm = outputs.get('feature_masks', {})
        Bn = labels.shape[0]
        # validity vectors depend on the masks alone: derived once per mask set (see _fusion)
        ckey = (Bn,) + tuple((m, id(t)) for m, t in fm.items())
        cent = self._loss_cache.get(ckey)
        if cent is None or any(a is not b for a, b in zip(cent['masks'], fm.values())):
            if fm:
                anyv = (torch.stack([t.to(dev) for t in fm.values()], dim=0) > 0).any(dim=0)
            else:
                anyv = torch.ones(Bn, dtype=torch.bool, device=dev)
            cent = dict(masks=list(fm.values()), valid=anyv.to(torch.uint8).contiguous())
            if 'vis' in fm:
                cent['gv'] = (fm['vis'] > 0).to(torch.uint8).contiguous()
            if len(self._loss_cache) > 16:
                self._loss_cache.clear()
            self._loss_cache[ckey] = cent
        valid = cent['valid']
        ce, cnt = CrossEntropyLSFn.apply(logits, labels, valid, 0.1)
        zero = torch.zeros((), device=dev)
        sdm = zero
        use_sdm = (self.current_epoch >= self.config.sdm_weight_warmup_epochs) and (self.contrastive_weight > 0)
        raw = outputs.get('raw_modality_features', {})
        if use_sdm and 'vis' in raw and 'vis' in fm:
            gv = cent['gv']
            mods = [m for m in raw if m != 'vis' and m in fm]
            if mods:
                # every non-vis modality against vis in ONE fused launch (csrc/sdm.hip): the query sides are stacked
                q = torch.stack([raw[m] for m in mods], dim=0)                                   # [P, B, D]
                qv = cent.get(('qv', tuple(mods)))
                if qv is None:
                    qv = cent[('qv', tuple(mods))] = torch.stack([(fm[m] > 0) for m in mods], dim=0).to(torch.uint8).contiguous()
                L, flag = SDMFn.apply(q, raw['vis'], labels, labels, qv, gv, float(self.sdm_temperature))
                sdm = L.sum() / flag.sum().clamp_min(1.0)                                        # mean over the pairs that contribute (model.py:617-622)
        total = self.ce_weight * ce + self.contrastive_weight * sdm
        return {'total_loss': total, 'ce_loss': ce, 'sdm_loss': sdm, 'contrastive_loss': sdm, 'ce_valid_cnt': LazyCount(cnt)}

    # ------------------------------------------------------------------ optimiser groups
    def get_learnable_params(self) -> List[Dict[str, Any]]:
        """models/model.py:661-729 + clip_backbone.py:342-371 (same group names and learning rates)."""
        c = self.config
        groups = OrderedDict((n, []) for n in ('clip_backbone', 'mer_loras', 'tokenizers', 'projections',
                                               'classification_head', 'other_modules'))
        for name, p in self.named_parameters():
            if name.startswith('clip_encoder.'):
                if 'clip_model' in name and p.requires_grad:
                    groups['clip_backbone'].append(p)
                elif 'lora' in name.lower():
                    groups['mer_loras'].append(p)
                elif 'patch_embeds' in name:
                    groups['tokenizers'].append(p)
                elif 'vision_proj' in name or 'text_proj' in name:
                    groups['projections'].append(p)
                else:
                    groups['tokenizers'].append(p)
            elif name.startswith('bn_neck.classifier'):
                groups['classification_head'].append(p)
            else:
                groups['other_modules'].append(p)
        lrs = {'clip_backbone': getattr(c, 'base_learning_rate', 1e-5), 'mer_loras': getattr(c, 'mer_learning_rate', 5e-5),
               'tokenizers': getattr(c, 'tokenizer_learning_rate', 5e-5), 'projections': getattr(c, 'fusion_learning_rate', 5e-5),
               'classification_head': 3e-3, 'other_modules': getattr(c, 'fusion_learning_rate', 5e-5)}
        return [{'params': ps, 'lr': lrs[n], 'name': n} for n, ps in groups.items() if ps]


def apply_reference_freeze(model: CLIPBasedMultiModalReIDModel):
    """train.py:1418-1425: train only names containing loras / bn_neck / null_tokens."""
    for name, p in model.named_parameters():
        p.requires_grad = ('loras' in name or 'feature_mixture' in name or 'bn_neck' in name or 'null_tokens' in name)
