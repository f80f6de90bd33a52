"""CPU oracle for the hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.  The product path (``prcv2025reid_amd``) never does;
it fails loudly if its HIP extension is missing.

What this is: a plain-PyTorch fp32, CPU, functional restatement (own code, no
``nn.Module``, no copied source) of the reference's algorithm for every row of
SURVEY.md section 8(a).  Each function cites the reference file:line it follows.
Weights come in as a flat ``state`` dict whose keys equal the reference's
``state_dict()`` keys (prcv2025reid_amd/weights.py).

Parity status: PINNED.  tests/golden/make_golden.py (run in the build container
only) imports the reference's own ``models/*`` with a locally constructed
random-init ``transformers.CLIPModel`` in place of the hub download, fills both
sides from the same seeded weights and stores the reference's outputs, losses
and gradients as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
this file against them (and against the reference's one known answer,
``sdm_loss._quick_check`` = 3.3439764976501465, models/sdm_loss.py:153-167).
The reference has no test-suite of its own for this path.

All stochastic regularisers of the reference (DropPath clip_backbone.py:126-142,
dropouts model.py:31-36,43,92-106,200, batch-level modality dropout
model.py:435-473) are OFF here unless their random outcome is passed in explicitly
(``drop_scales``, ``moddrop_keep``): parity is defined with them disabled or fixed.
"""
import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

LN_EPS = 1e-5           # clip_backbone.py:35-36,210; HF text layer_norm_eps
BN_EPS = 1e-5           # nn.BatchNorm1d default, model.py:196
BN_MOMENTUM = 0.1
FEAT_SCALE = 8.0        # model.py:219
LABEL_SMOOTHING = 0.1   # model.py:290


# --------------------------------------------------------------------------- basic ops
def layer_norm(x, w, b, eps=LN_EPS):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    """nn.GELU() default = exact erf form (mer_lora.py:257, model.py:103)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def quick_gelu(x):
    """HF CLIP text MLP activation ``quick_gelu``."""
    return x * torch.sigmoid(1.702 * x)


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def mer_linear(x, state, prefix, modality, scaling):
    """MERLinear.forward, mer_lora.py:80-99 with LoRAAdapter.forward :40-49.

    shared(x) + B_m(A_m(x)) * (alpha / r); adapter picked by modality name.
    """
    y = linear(x, state[prefix + '.shared_linear.weight'], state[prefix + '.shared_linear.bias'])
    a = state[f'{prefix}.loras.{modality}.lora_A.weight']
    bm = state[f'{prefix}.loras.{modality}.lora_B.weight']
    return y + ((x @ a.t()) @ bm.t()) * scaling


def attention_core(q, k, v, heads, mask=None):
    """softmax(QK^T / sqrt(hd) [+mask]) V over ``heads`` heads.

    q,k,v: [B, S, d].  ``mask``: additive, broadcastable to [B, heads, S, S].
    Follows the SDPA call at mer_lora.py:185-190 (scale head_dim**-0.5, :128-129).
    """
    B, S, d = q.shape
    hd = d // heads
    qh = q.view(B, S, heads, hd).transpose(1, 2)
    kh = k.view(B, k.shape[1], heads, hd).transpose(1, 2)
    vh = v.view(B, v.shape[1], heads, hd).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * (hd ** -0.5)
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    o = p @ vh
    return o.transpose(1, 2).reshape(B, S, d)


# --------------------------------------------------------------------------- A1 patch embed
def patch_embed(images, state, modality, patch):
    """PatchEmbed.forward, patch_embeds.py:45-76.

    3-channel input to a 1-channel embed (nir, sk) is channel-averaged first
    (:63-65).  Conv k=s=patch == GEMM over non-overlapping patches; output
    [B, n_patches, d] in row-major patch order.
    """
    w = state[f'clip_encoder.patch_embeds.{modality}.proj.weight']
    b = state[f'clip_encoder.patch_embeds.{modality}.proj.bias']
    cin = w.shape[1]
    x = images
    if x.shape[1] != cin:
        if x.shape[1] == 3 and cin == 1:
            x = x.mean(dim=1, keepdim=True)
        else:
            raise ValueError(f'channel mismatch {x.shape[1]} vs {cin}')
    B, C, H, W = x.shape
    gh, gw = H // patch, W // patch
    cols = x.view(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * patch * patch)
    return cols @ w.reshape(w.shape[0], -1).t() + b


# --------------------------------------------------------------------------- A2-A6 vision encoder
def vision_block(x, state, prefix, modality, heads, scaling, drop_scales=None):
    """MERTransformerBlock.forward, clip_backbone.py:61-85 (pre-LN, two residuals);
    attention = MERMultiheadAttention.forward mer_lora.py:141-231; MLP = MERMLP.forward
    mer_lora.py:267-280 (erf GELU).  ``drop_scales`` = (s_attn [B], s_mlp [B]): the per-sample DropPath factors
    ``floor(keep + U) / keep`` of clip_backbone.py:137-141, given explicitly so a test can fix the random draw."""
    h = layer_norm(x, state[prefix + '.ln1.weight'], state[prefix + '.ln1.bias'])
    q = mer_linear(h, state, prefix + '.attn.q_proj', modality, scaling)
    k = mer_linear(h, state, prefix + '.attn.k_proj', modality, scaling)
    v = mer_linear(h, state, prefix + '.attn.v_proj', modality, scaling)
    a = attention_core(q, k, v, heads)
    br = mer_linear(a, state, prefix + '.attn.out_proj', modality, scaling)
    if drop_scales is not None and drop_scales[0] is not None:
        br = br * drop_scales[0].view(-1, 1, 1)
    x = x + br
    h2 = layer_norm(x, state[prefix + '.ln2.weight'], state[prefix + '.ln2.bias'])
    u = mer_linear(h2, state, prefix + '.mlp.fc1', modality, scaling)
    br = mer_linear(gelu_erf(u), state, prefix + '.mlp.fc2', modality, scaling)
    if drop_scales is not None and drop_scales[1] is not None:
        br = br * drop_scales[1].view(-1, 1, 1)
    return x + br


def encode_vision(images, modality, state, arch, drop_scales=None):
    """CLIPUnifiedEncoder.encode_vision, clip_backbone.py:254-286.

    CLS + patches, + pos-embed, L blocks, final LN, CLS row, vision_proj (no bias).
    No pre-LN (the reference omits CLIP's pre_layrnorm)."""
    scaling = arch['lora_alpha'] / arch['lora_rank']
    pe = patch_embed(images, state, modality, arch['patch_size'])
    B = pe.shape[0]
    cls = state['clip_encoder.cls_token'].expand(B, -1, -1)
    x = torch.cat([cls, pe], dim=1) + state['clip_encoder.vision_pos_embed'].unsqueeze(0)
    for i in range(arch['vision_layers']):
        x = vision_block(x, state, f'clip_encoder.vision_layers.{i}', modality, arch['vision_heads'], scaling,
                         None if drop_scales is None else drop_scales[i])
    x0 = layer_norm(x[:, 0], state['clip_encoder.vision_ln_final.weight'], state['clip_encoder.vision_ln_final.bias'])
    return x0 @ state['clip_encoder.vision_proj.weight'].t()


# --------------------------------------------------------------------------- A7 text encoder
def encode_text(input_ids, attention_mask, state, arch):
    """CLIPUnifiedEncoder.encode_text, clip_backbone.py:288-313, with the HF
    ``CLIPTextModel`` it calls (transformers 5.15 modeling_clip: token+position
    embeddings, L pre-LN layers with causal + key-padding mask and quick_gelu,
    final LN, pooled at the first EOS position), then text_proj (no bias)."""
    tp = 'clip_encoder.clip_model.text_model.'
    B, T = input_ids.shape
    x = state[tp + 'embeddings.token_embedding.weight'][input_ids] + \
        state[tp + 'embeddings.position_embedding.weight'][:T].unsqueeze(0)
    neg = torch.finfo(torch.float32).min
    causal = torch.full((T, T), neg).triu(1)
    mask = causal.view(1, 1, T, T).expand(B, 1, T, T).clone()
    if attention_mask is not None:
        mask = mask.masked_fill((attention_mask == 0).view(B, 1, 1, T), neg)
    heads = arch['text_heads']
    for i in range(arch['text_layers']):
        lp = f'{tp}encoder.layers.{i}.'
        h = layer_norm(x, state[lp + 'layer_norm1.weight'], state[lp + 'layer_norm1.bias'])
        q = linear(h, state[lp + 'self_attn.q_proj.weight'], state[lp + 'self_attn.q_proj.bias'])
        k = linear(h, state[lp + 'self_attn.k_proj.weight'], state[lp + 'self_attn.k_proj.bias'])
        v = linear(h, state[lp + 'self_attn.v_proj.weight'], state[lp + 'self_attn.v_proj.bias'])
        a = attention_core(q, k, v, heads, mask)
        x = x + linear(a, state[lp + 'self_attn.out_proj.weight'], state[lp + 'self_attn.out_proj.bias'])
        h2 = layer_norm(x, state[lp + 'layer_norm2.weight'], state[lp + 'layer_norm2.bias'])
        u = quick_gelu(linear(h2, state[lp + 'mlp.fc1.weight'], state[lp + 'mlp.fc1.bias']))
        x = x + linear(u, state[lp + 'mlp.fc2.weight'], state[lp + 'mlp.fc2.bias'])
    x = layer_norm(x, state[tp + 'final_layer_norm.weight'], state[tp + 'final_layer_norm.bias'])
    eos_pos = (input_ids == arch['text_eos_id']).int().argmax(dim=-1)
    pooled = x[torch.arange(B), eos_pos]
    return pooled @ state['clip_encoder.text_proj.weight'].t()


# --------------------------------------------------------------------------- A9 SDM module
def sdm_module(x, state):
    """SemanticDisentanglementModule.forward, model.py:57-77.

    MHA over a length-1 sequence: softmax over one key == 1, so the attention
    output is out_proj(v_proj(x)); then Linear-LN-ReLU-Linear."""
    D = x.shape[1]
    wv = state['sdm_module.semantic_attn.in_proj_weight'][2 * D:3 * D]
    bv = state['sdm_module.semantic_attn.in_proj_bias'][2 * D:3 * D]
    a = linear(linear(x, wv, bv), state['sdm_module.semantic_attn.out_proj.weight'],
               state['sdm_module.semantic_attn.out_proj.bias'])
    y = x + a
    y = linear(y, state['sdm_module.semantic_proj.0.weight'], state['sdm_module.semantic_proj.0.bias'])
    y = layer_norm(y, state['sdm_module.semantic_proj.1.weight'], state['sdm_module.semantic_proj.1.bias'])
    y = torch.relu(y)
    return linear(y, state['sdm_module.semantic_proj.4.weight'], state['sdm_module.semantic_proj.4.bias'])


# --------------------------------------------------------------------------- A10 fusion
def feature_fusion(features: List[torch.Tensor], masks: Optional[List[torch.Tensor]], state, heads):
    """FeatureFusion.forward, model.py:113-183."""
    if len(features) == 1:
        return features[0]
    x = torch.stack(features, dim=1)                      # [B, M, D]
    B, M, D = x.shape
    add_mask = None
    sm = None
    if masks is not None:
        sm = torch.stack(masks, dim=1)                    # [B, M]
        pad = ~sm.bool()
        dead = pad.all(dim=1)
        if bool(dead.any()):                              # model.py:141-149
            pad = pad.clone(); pad[dead, 0] = False
            live = x[~dead]
            gm = live.mean(dim=(0, 1)) if live.numel() > 0 else torch.zeros(D)
            x = x.clone(); x[dead, 0] = gm
        add_mask = torch.zeros(B, 1, 1, M).masked_fill(pad.view(B, 1, 1, M), float('-inf'))
    w = state['feature_fusion.multihead_attn.in_proj_weight']
    b = state['feature_fusion.multihead_attn.in_proj_bias']
    q = linear(x, w[:D], b[:D]); k = linear(x, w[D:2 * D], b[D:2 * D]); v = linear(x, w[2 * D:], b[2 * D:])
    a = attention_core(q, k, v, heads, add_mask)
    a = linear(a, state['feature_fusion.multihead_attn.out_proj.weight'],
               state['feature_fusion.multihead_attn.out_proj.bias'])
    y = layer_norm(x + a, state['feature_fusion.norm1.weight'], state['feature_fusion.norm1.bias'])
    m = layer_norm(y, state['feature_fusion.mlp.0.weight'], state['feature_fusion.mlp.0.bias'])
    m = gelu_erf(linear(m, state['feature_fusion.mlp.1.weight'], state['feature_fusion.mlp.1.bias']))
    m = linear(m, state['feature_fusion.mlp.4.weight'], state['feature_fusion.mlp.4.bias'])
    z = layer_norm(y + m, state['feature_fusion.norm2.weight'], state['feature_fusion.norm2.bias'])
    z = torch.nan_to_num(z, nan=0.0, posinf=1e4, neginf=-1e4)
    if sm is None:
        return z.mean(dim=1)
    cnt = sm.sum(dim=1, keepdim=True).float().clamp(min=1.0)
    return (z * sm.unsqueeze(-1).float()).sum(dim=1) / cnt


# --------------------------------------------------------------------------- A11 BN-neck
def bn_neck(x, state, training: bool):
    """BNNeck.forward, model.py:208-224: BatchNorm1d -> 8 * L2-normalise -> classifier.

    Returns (bn_features, logits, batch_mean, batch_var_biased); the last two are
    None in eval.  Running-stat update rule is torch's (momentum 0.1, unbiased var)."""
    w, b = state['bn_neck.bn.weight'], state['bn_neck.bn.bias']
    if training:
        mu = x.mean(0); var = ((x - mu) ** 2).mean(0)
    else:
        mu, var = state['bn_neck.bn.running_mean'], state['bn_neck.bn.running_var']
    y = (x - mu) / torch.sqrt(var + BN_EPS) * w + b
    n = y.norm(dim=1, keepdim=True).clamp_min(1e-12)
    f = y / n * FEAT_SCALE
    logits = f @ state['bn_neck.classifier.weight'].t()
    return f, logits, (mu if training else None), (var if training else None)


# --------------------------------------------------------------------------- A8 forward
def forward(state, arch, images: Optional[Dict[str, torch.Tensor]], tokens: Optional[Dict[str, torch.Tensor]],
            modality_masks: Optional[Dict[str, torch.Tensor]], training: bool,
            moddrop_keep: Optional[Dict[str, bool]] = None, min_modalities: int = 1):
    """CLIPBasedMultiModalReIDModel.forward, model.py:321-510 (regularisers off).

    ``moddrop_keep`` = {modality: keep?}: the outcome of the batch-level modality-dropout draws of model.py:449-453
    (``torch.rand(1).item() > p`` per non-'vis' modality), given explicitly so a test can fix them; None = no dropout.
    Restated from model.py:435-473: dropped modalities leave BOTH the fused list and ``feature_masks``; the draw is
    cancelled if fewer than ``min_modalities`` would remain or a sample would be left without any valid modality.

    ``tokens`` = {'input_ids','attention_mask'} (the reference tokenises List[str]
    on the host, clip_backbone.py:297-303).  Quirk kept: with no mask for a vision
    modality nothing is encoded and its mask becomes zeros (model.py:367,386-389).
    """
    vmods = [m for m in arch['modalities'] if m != 'text']
    B = None
    if images:
        B = next(iter(images.values())).shape[0]
    elif tokens is not None:
        B = tokens['input_ids'].shape[0]
    if B is None:
        raise ValueError('cannot determine batch size')
    raw, fmask = {}, {}
    if images:
        for m, img in images.items():
            if m not in vmods:
                continue
            mask = modality_masks.get(m) if modality_masks is not None else None
            null = state[f'null_tokens.{m}']
            if mask is not None and float(mask.sum()) > 0:
                idx = mask.bool()
                feats = encode_vision(img[idx], m, state, arch)
                full = null.expand(B, -1).clone()
                full[idx] = feats
            else:
                full = null.expand(B, -1)
                mask = torch.zeros(B)
            raw[m] = full; fmask[m] = mask
    if tokens is not None and tokens['input_ids'].shape[0] > 0:
        tmask = modality_masks.get('text') if modality_masks is not None else None
        tf = encode_text(tokens['input_ids'], tokens.get('attention_mask'), state, arch)
        if tmask is not None:
            bad = ~tmask.bool()
            if bool(bad.any()):
                tf = tf.clone(); tf[bad] = state['null_tokens.text'].expand(int(bad.sum()), -1)
        else:
            tmask = torch.ones(B)
        raw['text'] = tf; fmask['text'] = tmask
    if not raw:
        raise ValueError('no modality given')
    return head(raw, fmask, state, arch, training, moddrop_keep, min_modalities)


def head(raw: Dict[str, torch.Tensor], fmask: Dict[str, torch.Tensor], state, arch, training: bool,
         moddrop_keep: Optional[Dict[str, bool]] = None, min_modalities: int = 1):
    """Everything of forward() after the encoders (model.py:392-510): SDM module per modality (training only), batch-level
    modality dropout, fusion, BN-neck.  ``raw`` = per-modality [B, D] features with null tokens already filled in."""
    sem = {m: (sdm_module(f, state) if training else f) for m, f in raw.items()}
    if training and moddrop_keep is not None and len(sem) > min_modalities:
        kept = [m for m in sem if m == 'vis' or moddrop_keep.get(m, True)]
        if len(kept) >= min_modalities:
            alive = torch.stack([fmask[m] for m in kept], dim=1).bool().any(dim=1)
            if bool(alive.all()):
                sem = {m: sem[m] for m in kept}
                fmask = {m: fmask[m] for m in kept}
    flist = list(sem.values()); mlist = [fmask[m] for m in sem]
    fused = flist[0] if len(flist) == 1 else feature_fusion(flist, mlist, state, arch['fusion_num_heads'])
    out = {'features': fused, 'raw_modality_features': raw, 'modality_features': sem, 'feature_masks': fmask}
    if 'bn_neck.classifier.weight' in state:
        f, logits, mu, var = bn_neck(fused, state, training)
        out['bn_features'] = f; out['logits'] = logits
        out['bn_batch_mean'] = mu; out['bn_batch_var'] = var
    return out


# --------------------------------------------------------------------------- A13 SDM loss
def _one_side(S, y):
    """_one_side_ce, sdm_loss.py:34-70: rows with >=1 positive; target = uniform over
    positives; mean over those rows of -sum q log_softmax(S)."""
    valid = y.sum(dim=1) > 0
    if not bool(valid.any()):
        return torch.zeros((), dtype=S.dtype)
    Sv = S[valid].clamp(-20.0, 20.0)
    pos = (y[valid] > 0).float()
    q = pos / pos.sum(dim=1, keepdim=True).clamp_min(1.0)
    return (-(q * torch.log_softmax(Sv, dim=1)).sum(dim=1)).mean()


def sdm_loss(qry, gal, y, tau=0.2, eps=1e-8):
    """sdm_loss_stable, sdm_loss.py:13-149: tau clamped to [0.15, 0.5] (:28), both
    sides L2-normalised with eps 1e-8 (:31-32), S = q g^T / tau (:86) clamped to +-20
    (:94), symmetric 0.5*(q2g + g2q) (:121-123); 0 if no row has a positive (:105-106)."""
    t = max(0.15, min(0.5, tau))
    qn = qry / qry.norm(dim=1, keepdim=True).clamp_min(eps)
    gn = gal / gal.norm(dim=1, keepdim=True).clamp_min(eps)
    S = (qn.float() @ gn.float().t() / t).clamp(-20.0, 20.0)
    if not bool((y.sum(dim=1) > 0).any()):
        return torch.zeros((), dtype=qry.dtype)
    return 0.5 * (_one_side(S, y) + _one_side(S.t(), y.t()))


# --------------------------------------------------------------------------- A12 total loss
def cross_entropy_ls(logits, labels, eps=LABEL_SMOOTHING):
    """nn.CrossEntropyLoss(label_smoothing=0.1), model.py:290: mean over rows of
    (1-eps)*nll + eps * mean_c(-log p_c)."""
    lp = torch.log_softmax(logits, dim=1)
    nll = -lp.gather(1, labels.view(-1, 1)).squeeze(1)
    smooth = -lp.mean(dim=1)
    return ((1.0 - eps) * nll + eps * smooth).mean()


def compute_loss(out, labels, *, ce_weight=1.0, contrastive_weight=0.1, tau=0.2, use_sdm=True):
    """CLIPBasedMultiModalReIDModel.compute_loss, model.py:512-659.

    ``use_sdm`` stands for ``current_epoch >= sdm_weight_warmup_epochs and
    contrastive_weight > 0`` (:552)."""
    logits = out['logits']; fm = out['feature_masks']
    anyv = torch.zeros(labels.shape[0], dtype=torch.bool)
    if fm:
        for m in fm.values():
            anyv |= (m > 0)
    else:
        anyv[:] = True
    ok = anyv & (labels >= 0) & (labels < logits.shape[1])
    cnt = int(ok.sum())
    ce = cross_entropy_ls(logits[ok], labels[ok]) if cnt > 0 else torch.zeros(())
    sdm = torch.zeros(())
    if use_sdm and contrastive_weight > 0:
        raw = out['raw_modality_features']
        if 'vis' in raw and 'vis' in fm and float((fm['vis'] > 0).sum()) > 0:
            vi = fm['vis'] > 0
            vfeat, vlab = raw['vis'][vi], labels[vi]
            parts = []
            for m, feat in raw.items():
                if m == 'vis' or m not in fm:
                    continue
                mi = fm[m] > 0
                if int(mi.sum()) == 0:
                    continue
                y = (labels[mi].view(-1, 1) == vlab.view(1, -1)).float()
                if y.numel() == 0 or float(y.sum()) == 0:
                    continue
                L = sdm_loss(feat[mi], vfeat, y, tau)
                if bool(torch.isfinite(L)):
                    parts.append(L)
            if parts:
                sdm = torch.stack(parts).mean()
    total = ce_weight * ce + contrastive_weight * sdm
    return {'total_loss': total, 'ce_loss': ce, 'sdm_loss': sdm, 'contrastive_loss': sdm, 'ce_valid_cnt': cnt}


# --------------------------------------------------------------------------- R1 / R2 retrieval
def l2n(x, eps=1e-12):
    """F.normalize(x, dim=-1) (train.py:442, eval_mm_protocol.py:45-47)."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def cosine_sim(a, b):
    """eval_mm_protocol.py:50-53 / train.py:499: a @ b.T on L2-normalised rows."""
    return a @ b.t()


def rank_full(sim_row):
    """Full descending ranking of one query (train.py:463, eval_mm_protocol.py:423).
    The reference's argsort is unstable; the contract here is (score desc, index asc)."""
    return torch.argsort(sim_row, descending=True, stable=True)


def topk_ranklist(Q, G, k, exclude: Optional[torch.Tensor] = None):
    """First k entries of rank_full for every query; ``exclude[q, g]`` True sets the
    score to -1e9 before ranking (same-image mask, eval_mm_protocol.py:421-422)."""
    sim = cosine_sim(Q, G)
    if exclude is not None:
        sim = sim.masked_fill(exclude, -1e9)
    idx = torch.argsort(sim, dim=1, descending=True, stable=True)[:, :k]
    return idx, sim.gather(1, idx)


def reid_map(sim, q_ids, g_ids):
    """_reid_map, train.py:450-479: AP over the full ranking; sum(AP) / #queries that
    have a positive; top-1 / Nq."""
    Nq, Ng = sim.shape
    ar = torch.arange(1, Ng + 1, dtype=torch.float32)
    ap_sum, top1 = 0.0, 0.0
    for i in range(Nq):
        order = rank_full(sim[i])
        hit = (g_ids[order] == q_ids[i]).float()
        rel = float(hit.sum())
        if rel == 0:
            continue
        ap_sum += float(((torch.cumsum(hit, 0) / ar) * hit).sum() / rel)
        top1 += float(hit[0])
    valid = max(1, int((q_ids.view(-1, 1) == g_ids.view(1, -1)).any(dim=1).sum()))
    return ap_sum / valid, top1 / Nq


def rank_and_metrics(q_feats, q_pids, g_feats, g_pids, q_imgids: Optional[Sequence] = None,
                     g_imgids: Optional[Sequence] = None):
    """Metric half of rank_and_metrics, eval_mm_protocol.py:401-469: per query
    sims = q @ G^T, same-img entries masked to -1e9, CMC@1/5/10 from the first ten
    ranks, AP over the full ranking (early exit once all positives are found);
    queries with no positive in the gallery are skipped."""
    APs, h1, h5, h10 = [], [], [], []
    for i in range(q_feats.shape[0]):
        sims = cosine_sim(q_feats[i:i + 1], g_feats).squeeze(0)
        keep = torch.ones_like(sims, dtype=torch.bool)
        if q_imgids is not None and g_imgids is not None:
            qs = q_imgids[i] if isinstance(q_imgids[i], (set, list, tuple)) else {q_imgids[i]}
            qs = {x for x in qs if x is not None}
            if qs:
                keep = torch.tensor([g not in qs for g in g_imgids], dtype=torch.bool)
        sm = sims.clone(); sm[~keep] = -1e9
        ranks = rank_full(sm)
        pos = ((g_pids == q_pids[i]) & keep)
        npos = int(pos.sum())
        if npos == 0:
            continue
        top = ranks[:10]
        h1.append(int(bool(pos[top[:1]].any()))); h5.append(int(bool(pos[top[:5]].any())))
        h10.append(int(bool(pos[top[:10]].any())))
        hit = pos[ranks].float()
        prec = torch.cumsum(hit, 0) / torch.arange(1, hit.numel() + 1, dtype=torch.float32)
        APs.append(float((prec * hit).sum() / npos))
    mean = lambda v: float(sum(v) / len(v)) if v else 0.0
    return {'mAP': mean(APs), 'R@1': mean(h1), 'R@5': mean(h5), 'R@10': mean(h10), 'num_queries': len(APs)}
