"""Configuration surface of the hot path.

Mirrors the field names and defaults of the reference's ``TrainingConfig``
(reference: configs/config.py:7-173) so a caller written against the reference
(``train.py``, ``tools/eval_mm_protocol.py``) can hand the same object to
:class:`prcv2025reid_amd.model.CLIPBasedMultiModalReIDModel`.

Differences, on purpose:
  * no torchvision import and no directory creation in ``__post_init__``
    (reference: configs/config.py:175-185) -- those belong to the data side,
    which is out of scope for this path;
  * architecture knobs for the backbone (``vision_layers`` ... ``text_vocab``)
    are explicit fields here, because the reference takes them from the
    downloaded HF checkpoint (models/clip_backbone.py:170,195-206), which is
    not available offline.  Defaults are CLIP ViT-B/16.

The model reads every field with ``getattr(config, name, default)`` exactly as
the reference does (models/model.py:237-294,316,436-440,552), so any plain
object with these attributes works.
"""
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class TrainingConfig:
    # data (unused by the hot path; kept for surface compatibility)
    data_root: str = "./data/train"
    json_file: str = "./data/train/text_annos.json"
    val_ratio: float = 0.2
    seed: int = 42

    # backbone
    clip_model_name: str = "openai/clip-vit-base-patch16"
    use_clip_backbone: bool = True
    fusion_dim: int = 512
    vision_hidden_dim: int = 768

    # MER LoRA
    enable_mer: bool = True
    mer_lora_rank: int = 4
    mer_lora_alpha: float = 1.0
    modalities: List[str] = field(default_factory=lambda: ['vis', 'nir', 'sk', 'cp', 'text'])
    patch_size: int = 16
    freeze_text_backbone: bool = False
    drop_path: float = 0.15
    image_size: int = 224
    feature_dim: int = 2048
    hidden_dim: int = 512
    dropout_rate: float = 0.5

    # P x K
    num_ids_per_batch: int = 3
    instances_per_id: int = 2
    allow_id_reuse: bool = True
    sampling_fallback: bool = True
    min_modal_coverage: float = 0.8
    force_modal_pairs: bool = True
    gradient_accumulation_steps: int = 1
    freeze_backbone: bool = True
    num_epochs: int = 60

    # learning rates
    base_learning_rate: float = 5e-6
    mer_learning_rate: float = 2e-5
    tokenizer_learning_rate: float = 2e-5
    fusion_learning_rate: float = 2e-5
    head_learning_rate: float = 3e-3
    head_lr_warmup_epochs: int = 2
    weight_decay: float = 1e-4
    warmup_epochs: int = 5
    scheduler: str = "cosine"
    conservative_factor: float = 0.7
    adaptive_gradient_clip: bool = True
    stability_monitoring: bool = True

    # losses
    ce_weight: float = 1.0
    sdm_weight_warmup_epochs: int = 1
    sdm_weight_schedule: List[float] = field(default_factory=lambda: [0.1, 0.3, 0.5])
    sdm_weight_initial: float = 0.1
    sdm_weight_final: float = 0.5
    sdm_weight_max: float = 0.5
    contrastive_weight: float = 0.0
    sdm_semantic_dim: int = 512
    sdm_num_heads: int = 8
    sdm_dropout: float = 0.1   # not a reference config field: hard-coded in SemanticDisentanglementModule (models/model.py:35,43)
    sdm_temperature: float = 0.2
    sdm_init_temperature: float = 0.18
    sdm_final_temperature: float = 0.16
    sdm_fallback_temperature: float = 0.20
    sdm_learnable_temp: bool = True
    sdm_temp_warmup_epochs: int = 3

    # fusion
    fusion_num_heads: int = 8
    fusion_mlp_ratio: float = 2.0
    fusion_dropout: float = 0.1

    # augmentation / modality dropout
    random_flip: bool = True
    random_crop: bool = True
    color_jitter: bool = True
    random_erase: float = 0.3
    modality_dropout: float = 0.15
    modality_dropout_warmup_epochs: int = 3
    min_modalities: int = 1
    require_modal_pairs: bool = True
    modal_pair_retry_limit: int = 3
    modal_pair_fallback_ratio: float = 0.3
    pair_coverage_target: float = 0.85
    pair_coverage_window: int = 100

    # device / loader
    device: str = "cuda"
    num_workers: int = 2
    pin_memory: bool = True
    persistent_workers: bool = True
    prefetch_factor: int = 2

    # logging / eval
    save_dir: str = "./checkpoints"
    log_dir: str = "./logs"
    save_freq: int = 20
    eval_freq: int = 15
    eval_sample_ratio: float = 0.3
    eval_include_patterns: List[str] = field(default_factory=lambda: [
        "single/nir", "single/sk", "single/cp", "single/text", "quad/nir+sk+cp+text"])
    eval_every_n_epoch: int = 1
    eval_every_n_steps: int = 0
    do_eval: bool = True
    eval_after_steps: Optional[int] = None
    eval_cache_dir: str = "./.eval_cache"
    eval_cache_tag: str = "val_v1"
    inference_batch_size: int = 8
    best_model_path: str = "./checkpoints/best_model.pth"
    label_smoothing: float = 0.1

    # 16-bit MFMA operand format of the HIP path: 'bf16' (8 significant bits) or 'f16' (11 bits, backward pass
    # runs loss-scaled).  None -> environment REID_T16, default 'bf16'.  See DESIGN.md "Precision".
    compute_dtype: Optional[str] = None
    # Initial values of a freshly constructed model: 'reference' = the reference's construction semantics (lora_B = 0, lora_A
    # kaiming-uniform, xavier SDM module, CLIP-derived patch convolutions + noise; seeded stand-ins where the reference downloads
    # CLIP), 'seeded' = every tensor random (prcv2025reid_amd.weights.seeded_state: what the parity fixtures and the bench use).
    init: str = 'reference'

    # ---- backbone architecture (taken from the HF checkpoint in the reference) ----
    vision_layers: int = 12
    vision_heads: int = 12
    vision_mlp_dim: int = 3072
    text_hidden_dim: int = 512
    text_layers: int = 12
    text_heads: int = 8
    text_mlp_dim: int = 2048
    text_vocab: int = 49408
    text_max_len: int = 77
    text_eos_id: int = 49407
    text_bos_id: int = 49406


def arch_of(config) -> dict:
    """Backbone architecture numbers of ``config`` (CLIP ViT-B/16 defaults)."""
    d = TrainingConfig()
    g = lambda n: getattr(config, n, getattr(d, n))
    return dict(
        modalities=list(g('modalities')),
        fusion_dim=g('fusion_dim'), vision_hidden_dim=g('vision_hidden_dim'),
        image_size=g('image_size'), patch_size=g('patch_size'),
        vision_layers=g('vision_layers'), vision_heads=g('vision_heads'),
        vision_mlp_dim=g('vision_mlp_dim'),
        text_hidden_dim=g('text_hidden_dim'), text_layers=g('text_layers'),
        text_heads=g('text_heads'), text_mlp_dim=g('text_mlp_dim'),
        text_vocab=g('text_vocab'), text_max_len=g('text_max_len'),
        text_eos_id=g('text_eos_id'), text_bos_id=g('text_bos_id'),
        lora_rank=getattr(config, 'mer_lora_rank', 4),
        lora_alpha=getattr(config, 'mer_lora_alpha', 1.0),
        sdm_semantic_dim=g('sdm_semantic_dim'), sdm_num_heads=g('sdm_num_heads'),
        fusion_num_heads=g('fusion_num_heads'), fusion_mlp_ratio=g('fusion_mlp_ratio'),
    )
