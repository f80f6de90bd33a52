"""Parameter inventory of the hot path and a seeded, order-independent initialiser.

Key names equal the reference's ``state_dict()`` keys (SURVEY.md section 8b;
modules at models/model.py:255-303,313, models/clip_backbone.py:177-219,
models/mer_lora.py:30-31,72-78, models/patch_embeds.py:30-39,122-125) so a
reference checkpoint's hot-path tensors load by name.

The reference fetches pretrained CLIP weights by model NAME
(models/clip_backbone.py:170), which is impossible offline, so every parity
fixture and the benchmark use :func:`seeded_fill`: each tensor is drawn from its
own ``torch.Generator`` seeded by (seed, crc32(key)).  That makes the values
independent of iteration order and of which other keys exist, so the reference
model (which also carries unused tensors such as ``clip_model.vision_model.*``)
and this package produce identical hot-path weights from the same seed.
"""
import re
import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import torch

VISION_MODALITIES_DEFAULT = ['vis', 'nir', 'sk', 'cp']
PATCH_CHANNELS = {'vis': 3, 'nir': 1, 'cp': 3, 'sk': 1}  # models/patch_embeds.py:122-125


def param_spec(arch: dict, num_classes=None) -> "OrderedDict[str, Tuple[int, ...]]":
    """Hot-path tensors: name -> shape.  ``arch`` comes from config.arch_of()."""
    d = arch['vision_hidden_dim']; ff = arch['vision_mlp_dim']; D = arch['fusion_dim']
    r = arch['lora_rank']
    P = arch['patch_size']; n_tok = (arch['image_size'] // P) ** 2 + 1
    td = arch['text_hidden_dim']; tff = arch['text_mlp_dim']
    vmods = [m for m in arch['modalities'] if m != 'text']
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    ce = 'clip_encoder.'
    s[ce + 'vision_pos_embed'] = (n_tok, d)
    s[ce + 'cls_token'] = (1, 1, d)
    # text tower (HF CLIPTextModel layout under clip_model.text_model)
    tp = ce + 'clip_model.text_model.'
    s[tp + 'embeddings.token_embedding.weight'] = (arch['text_vocab'], td)
    s[tp + 'embeddings.position_embedding.weight'] = (arch['text_max_len'], td)
    for i in range(arch['text_layers']):
        lp = f'{tp}encoder.layers.{i}.'
        for nm in ('k_proj', 'v_proj', 'q_proj', 'out_proj'):
            s[f'{lp}self_attn.{nm}.weight'] = (td, td)
            s[f'{lp}self_attn.{nm}.bias'] = (td,)
        s[lp + 'layer_norm1.weight'] = (td,); s[lp + 'layer_norm1.bias'] = (td,)
        s[lp + 'mlp.fc1.weight'] = (tff, td); s[lp + 'mlp.fc1.bias'] = (tff,)
        s[lp + 'mlp.fc2.weight'] = (td, tff); s[lp + 'mlp.fc2.bias'] = (td,)
        s[lp + 'layer_norm2.weight'] = (td,); s[lp + 'layer_norm2.bias'] = (td,)
    s[tp + 'final_layer_norm.weight'] = (td,); s[tp + 'final_layer_norm.bias'] = (td,)
    # per-modality patch embeds (no sharing)
    for m in ('vis', 'nir', 'cp', 'sk'):
        c = PATCH_CHANNELS[m]
        s[f'{ce}patch_embeds.{m}.proj.weight'] = (d, c, P, P)
        s[f'{ce}patch_embeds.{m}.proj.bias'] = (d,)
    # MER ViT blocks
    for i in range(arch['vision_layers']):
        lp = f'{ce}vision_layers.{i}.'
        s[lp + 'ln1.weight'] = (d,); s[lp + 'ln1.bias'] = (d,)
        s[lp + 'ln2.weight'] = (d,); s[lp + 'ln2.bias'] = (d,)
        lin = [('attn.q_proj', d, d), ('attn.k_proj', d, d), ('attn.v_proj', d, d),
               ('attn.out_proj', d, d), ('mlp.fc1', d, ff), ('mlp.fc2', ff, d)]
        for nm, fin, fout in lin:
            s[f'{lp}{nm}.shared_linear.weight'] = (fout, fin)
            s[f'{lp}{nm}.shared_linear.bias'] = (fout,)
            for m in vmods:
                s[f'{lp}{nm}.loras.{m}.lora_A.weight'] = (r, fin)
                s[f'{lp}{nm}.loras.{m}.lora_B.weight'] = (fout, r)
    s[ce + 'vision_ln_final.weight'] = (d,); s[ce + 'vision_ln_final.bias'] = (d,)
    s[ce + 'vision_proj.weight'] = (D, d)
    s[ce + 'text_proj.weight'] = (D, td)
    # SDM module (nn.MultiheadAttention + Sequential[Linear, LN, ReLU, Dropout, Linear])
    sd = arch['sdm_semantic_dim']
    s['sdm_module.semantic_attn.in_proj_weight'] = (3 * D, D)
    s['sdm_module.semantic_attn.in_proj_bias'] = (3 * D,)
    s['sdm_module.semantic_attn.out_proj.weight'] = (D, D)
    s['sdm_module.semantic_attn.out_proj.bias'] = (D,)
    s['sdm_module.semantic_proj.0.weight'] = (sd, D); s['sdm_module.semantic_proj.0.bias'] = (sd,)
    s['sdm_module.semantic_proj.1.weight'] = (sd,); s['sdm_module.semantic_proj.1.bias'] = (sd,)
    s['sdm_module.semantic_proj.4.weight'] = (sd, sd); s['sdm_module.semantic_proj.4.bias'] = (sd,)
    # fusion
    fh = int(D * arch['fusion_mlp_ratio'])
    s['feature_fusion.multihead_attn.in_proj_weight'] = (3 * D, D)
    s['feature_fusion.multihead_attn.in_proj_bias'] = (3 * D,)
    s['feature_fusion.multihead_attn.out_proj.weight'] = (D, D)
    s['feature_fusion.multihead_attn.out_proj.bias'] = (D,)
    s['feature_fusion.mlp.0.weight'] = (D,); s['feature_fusion.mlp.0.bias'] = (D,)
    s['feature_fusion.mlp.1.weight'] = (fh, D); s['feature_fusion.mlp.1.bias'] = (fh,)
    s['feature_fusion.mlp.4.weight'] = (D, fh); s['feature_fusion.mlp.4.bias'] = (D,)
    s['feature_fusion.norm1.weight'] = (D,); s['feature_fusion.norm1.bias'] = (D,)
    s['feature_fusion.norm2.weight'] = (D,); s['feature_fusion.norm2.bias'] = (D,)
    for m in arch['modalities']:
        s[f'null_tokens.{m}'] = (1, D)
    if num_classes is not None:
        s['bn_neck.bn.weight'] = (D,); s['bn_neck.bn.bias'] = (D,)
        s['bn_neck.bn.running_mean'] = (D,); s['bn_neck.bn.running_var'] = (D,)
        s['bn_neck.classifier.weight'] = (num_classes, D)
    return s


# keys a reference checkpoint carries that the hot path never reads
# (unused HF vision tower, dead channel_adapter, clones).
DEAD_KEY_PATTERNS = (
    r'^clip_encoder\.clip_model\.vision_model\.',
    r'^clip_encoder\.clip_model\.(logit_scale|visual_projection\.weight|text_projection\.weight)$',
    r'^clip_encoder\.patch_embeds\.(nir|sk)\.channel_adapter\.weight$',
    r'^bn_neck\.bn\.num_batches_tracked$',
)


def is_dead_key(key: str) -> bool:
    return any(re.search(p, key) for p in DEAD_KEY_PATTERNS)


def _rule(key: str):
    """(kind, scale) used by seeded_fill for ``key``."""
    if key.endswith('num_batches_tracked'):
        return 'zero', 0.0
    if key.endswith('running_var'):
        return 'var', 0.1
    if key.endswith('running_mean'):
        return 'normal', 0.1
    norm_w = (r'\.ln[12]\.weight$', r'layer_norm[12]\.weight$', r'layernorm\.weight$', r'layrnorm\.weight$',
              r'final_layer_norm\.weight$', r'vision_ln_final\.weight$', r'\.norm[12]\.weight$',
              r'bn_neck\.bn\.weight$', r'semantic_proj\.1\.weight$', r'feature_fusion\.mlp\.0\.weight$')
    if any(re.search(p, key) for p in norm_w):
        return 'one_plus', 0.1
    if key.endswith('.bias') or key.endswith('in_proj_bias'):
        return 'normal', 0.02
    if 'lora_A' in key:
        return 'normal', 0.05
    if 'lora_B' in key:
        return 'normal', 0.1
    if 'embedding' in key or key.endswith('vision_pos_embed') or key.endswith('cls_token'):
        return 'normal', 0.05
    if key.startswith('null_tokens.'):
        return 'normal', 0.02
    if key.endswith('logit_scale'):
        return 'const', 2.6592
    if 'sdm_module' in key or 'feature_fusion' in key:
        return 'normal', 0.04
    return 'normal', 0.02


def seeded_tensor(key: str, shape: Iterable[int], seed: int) -> torch.Tensor:
    """The fp32 CPU value of tensor ``key`` for ``seed`` (int64 for counters)."""
    kind, scale = _rule(key)
    shape = tuple(int(x) for x in shape)
    if kind == 'zero':
        return torch.zeros(shape, dtype=torch.int64)
    if kind == 'const':
        return torch.full(shape, scale, dtype=torch.float32)
    g = torch.Generator(device='cpu')
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    t = torch.randn(shape, generator=g, dtype=torch.float32)
    if kind == 'one_plus':
        return 1.0 + scale * t
    if kind == 'var':
        return 1.0 + scale * t.abs()
    return scale * t


@torch.no_grad()
def seeded_fill(named_tensors: Dict[str, torch.Tensor], seed: int) -> None:
    """In-place: set every tensor in ``named_tensors`` to its seeded value."""
    for k, t in named_tensors.items():
        v = seeded_tensor(k, t.shape, seed)
        t.copy_(v.to(dtype=t.dtype))


def seeded_state(arch: dict, num_classes, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """Fresh fp32 CPU hot-path state dict for ``seed``."""
    return OrderedDict((k, seeded_tensor(k, shp, seed)) for k, shp in param_spec(arch, num_classes).items())


# ---------------------------------------------------------------------------------------------------- construction semantics
def reference_init_state(arch: dict, num_classes, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """Fresh fp32 CPU state dict with the reference's CONSTRUCTION semantics (what a freshly built reference model holds):

    * tensors the reference takes from pretrained CLIP (text tower, the shared linears / LayerNorms of the MER blocks,
      position / class embeddings, the two projections, the 'vis' patch convolution) -- the hub is unreachable here, so they
      are the seeded stand-ins of :func:`seeded_tensor` (load real ones with checkpoint.load_clip_pretrained);
    * LoRA: ``lora_A`` kaiming-uniform(a=sqrt 5) = U(+-1/sqrt(fan_in)), ``lora_B`` ZERO, so the initial low-rank update is
      exactly 0 (mer_lora.py:36-38);
    * patch convolutions of the other modalities = the CLIP ('vis') one -- channel mean for the one-channel nir / sk -- plus
      N(0, 0.02^2) noise on the weight and N(0, 0.01^2) on the bias (patch_embeds.py:78-105,150-167);
    * SDM module: xavier-uniform Linear / in_proj weights, zero biases (model.py:50-55); fusion block: torch's defaults
      (xavier in_proj, kaiming-uniform(a=sqrt 5) Linears with U(+-1/sqrt(fan_in)) biases, zero MHA biases);
      LayerNorms (1, 0); null tokens N(0, 0.02^2) (model.py:300-303); BN-neck as models/model.py:196-206.
    Every draw comes from the per-key seeded generator, so the result does not depend on iteration order."""
    import math

    def gen(key):
        g = torch.Generator(device='cpu')
        g.manual_seed((int(seed) * 1000003 + zlib.crc32(('init:' + key).encode())) % (2 ** 63 - 1))
        return g

    def uniform(key, shape, bound):
        return (torch.rand(tuple(shape), generator=gen(key), dtype=torch.float32) * 2.0 - 1.0) * bound

    def normal(key, shape, std):
        return torch.randn(tuple(shape), generator=gen(key), dtype=torch.float32) * std

    out = OrderedDict()
    spec = param_spec(arch, num_classes)
    ce = 'clip_encoder.'
    vis_w = seeded_tensor(f'{ce}patch_embeds.vis.proj.weight', spec[f'{ce}patch_embeds.vis.proj.weight'], seed)
    for k, shp in spec.items():
        if '.lora_A.' in k:
            v = uniform(k, shp, 1.0 / math.sqrt(shp[1]))
        elif '.lora_B.' in k:
            v = torch.zeros(shp)
        elif k.startswith(ce + 'patch_embeds.'):
            m = k.split('.')[2]
            fan_in = spec[f'{ce}patch_embeds.{m}.proj.weight'][1] * shp[-1] * shp[-1] if k.endswith('weight') else None
            if k.endswith('weight'):
                base = vis_w if shp[1] == 3 else vis_w.mean(dim=1, keepdim=True)
                v = base.clone() if m == 'vis' else base + normal(k, shp, 0.02)
            else:                              # HF CLIP's patch convolution has no bias: each module keeps its own default draw
                w_shape = spec[f'{ce}patch_embeds.{m}.proj.weight']
                bound = 1.0 / math.sqrt(w_shape[1] * w_shape[2] * w_shape[3])
                v = uniform(k, shp, bound)
                if m != 'vis':
                    v = v + normal(k + ':noise', shp, 0.01)
        elif k.startswith('sdm_module.') or k.startswith('feature_fusion.'):
            ln = ('semantic_proj.1.', 'feature_fusion.mlp.0.', 'feature_fusion.norm1.', 'feature_fusion.norm2.')
            if any(t in k for t in ln):
                v = torch.ones(shp) if k.endswith('weight') else torch.zeros(shp)
            elif k.endswith('in_proj_weight') or (k.startswith('sdm_module.') and k.endswith('weight')):
                v = uniform(k, shp, math.sqrt(6.0 / (shp[0] + shp[1])))                   # xavier_uniform
            elif k.endswith('in_proj_bias') or k.endswith('out_proj.bias') or k.startswith('sdm_module.'):
                v = torch.zeros(shp)
            elif k.endswith('weight'):
                v = uniform(k, shp, 1.0 / math.sqrt(shp[1]))                              # nn.Linear default
            else:
                fan_in = spec[k[:-4] + 'weight'][1]
                v = uniform(k, shp, 1.0 / math.sqrt(fan_in))
        elif k.startswith('null_tokens.'):
            v = normal(k, shp, 0.02)
        elif k.startswith('bn_neck.'):
            if k.endswith('bn.weight') or k.endswith('running_var'):
                v = torch.ones(shp)
            elif k.endswith('bn.bias') or k.endswith('running_mean'):
                v = torch.zeros(shp)
            else:
                v = normal(k, shp, 0.001)
        else:
            v = seeded_tensor(k, shp, seed)
        out[k] = v
    return out


def fingerprint(state: Dict[str, torch.Tensor], keys=None) -> float:
    """Order-independent checksum used by fixtures to detect RNG drift."""
    tot = 0.0
    for k in (keys or sorted(state.keys())):
        tot += float(state[k].double().abs().sum()) * ((zlib.crc32(k.encode()) % 97) + 1)
    return tot
