"""GPU: the per-GPU slices of BASELINE.json configs[2] (DP: P=32,K=4, r=8, masks all-on) and configs[4] (stress: P=64,K=4, r=16,
Bernoulli(0.7) modality masks -- datasets/dataset.py:533-542 --, gradient accumulation 2 -- train.py:1364,1482-1483) at FULL
model dimensions, through the product step driver (train.py:684-1245 -> trainer.StepDriver).

The oracle's encoders need minutes of CPU at these batch sizes, so the checks are the size-independent ones:
  * head consistency: the oracle's head (SDM module, fusion, BN-neck, CE + SDM losses; oracle/reid_oracle.py head()) evaluated on
    the HIP model's OWN encoder outputs and masks reproduces the HIP model's fused feature, bn_features and three losses -- the
    whole head and loss stack at B = 128 / 256 with ragged masks (the encoders at full dimensions are pinned by
    test_config2_full_size_vs_oracle and the reference fixtures);
  * ce_valid_cnt = number of samples with at least one valid modality (model.py:540-546);
  * gradient accumulation: the gradient the optimizer sees after two micro-batches through StepDriver(accum_steps=2) equals
    (g_A + g_B) / 2 of the two micro-batches run separately (train.py:833-834,895), and the optimizer runs once;
  * everything finite, LoRA gradients non-zero.
The 8-GPU forms of both configs need a node (the driver's SCALE run); the arithmetic of the data-parallel path itself is
tests/test_parallel_gpu.py.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(rank, C, flavor='f16'):
    from prcv2025reid_amd.config import TrainingConfig
    from prcv2025reid_amd.model import CLIPBasedMultiModalReIDModel, apply_reference_freeze
    cfg = TrainingConfig(device='cuda', mer_lora_rank=rank, contrastive_weight=0.1, compute_dtype=flavor, init='seeded', seed=0,
                         drop_path=0.0, modality_dropout=0.0, dropout_rate=0.0, fusion_dropout=0.0, sdm_dropout=0.0)
    model = CLIPBasedMultiModalReIDModel(cfg)
    model.set_num_classes(C)
    apply_reference_freeze(model)
    model.set_epoch(2); model.train()
    return model, cfg


def _batch(model, P, K, seed, mask_drop, C):
    from prcv2025reid_amd.synthetic import synthetic_batch
    b = synthetic_batch(P, K, model.arch, seed=seed, mask_drop=mask_drop, num_classes=C)
    tok = model.tokenizer(b['texts'], return_tensors='pt', padding=True, truncation=True, max_length=77)
    return (dict(images={m: t.cuda() for m, t in b['images'].items()}, texts={k: v.cuda() for k, v in tok.items()},
                 modality_masks=b['modality_mask']), b['person_id'].cuda(), b)


def _head_consistency(model, cfg, out, L, labels_cpu):
    """oracle head on the HIP model's encoder outputs == the HIP head and losses (fp32 both; the HIP head kernels sum in another order)"""
    from oracle import reid_oracle as O
    state = {k: v.detach().cpu() for k, v in model.state_dict().items() if not k.startswith('clip_encoder.')}
    raw = {m: t.detach().cpu() for m, t in out['raw_modality_features'].items()}
    fm = {m: t.detach().cpu().float() for m, t in out['feature_masks'].items()}
    with torch.no_grad():
        ref = O.head(raw, fm, state, model.arch, True)
        Lr = O.compute_loss(ref, labels_cpu, contrastive_weight=0.1, tau=cfg.sdm_temperature)
    unit = lambda t: torch.nn.functional.normalize(t.detach().cpu().float(), dim=1)
    d_fused = float((unit(out['features']) - unit(ref['features'])).abs().max())
    d_bn = float((out['bn_features'].detach().cpu() / 8 - ref['bn_features'] / 8).abs().max())
    dl = {k: abs(float(L[k].detach()) - float(Lr[k])) for k in ('total_loss', 'ce_loss', 'sdm_loss')}
    print(f'  head consistency at B = {labels_cpu.shape[0]}: fused {d_fused:.2e}, bn/8 {d_bn:.2e}, losses {dl}')
    assert d_fused <= 2e-5 and d_bn <= 1e-4, (d_fused, d_bn)
    for k, v in dl.items():
        assert v <= 1e-4 * max(1.0, abs(float(Lr[k]))), (k, v)
    assert int(L['ce_valid_cnt']) == int(Lr['ce_valid_cnt'])
    return Lr


def test_config3_per_gpu_slice_full_size():
    """configs[2]: P=32,K=4 per GPU, r=8, masks all-on: one step of the product path on one rank's slice (DataParallel wrapper
    with a single-process world, the same object the N-GPU bench builds)."""
    from prcv2025reid_amd.parallel import DataParallel
    from prcv2025reid_amd.trainer import FusedAdamW, StepDriver
    C = 400
    model, cfg = _model(8, C)
    inp, labels, b = _batch(model, 32, 4, 1000, 0.0, C)
    dp = DataParallel(model)
    out = dp.forward(**inp)
    assert out['bn_features'].shape == (128, 512)
    L = dp.compute_loss(out, labels)
    _head_consistency(model, cfg, out, L, b['person_id'])
    assert int(L['ce_valid_cnt']) == 128
    L['total_loss'].backward()
    g = model.lora_arena.grad
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    for k, p in model.named_parameters():                   # masks all-on: no null token is ever used
        if k.startswith('null_tokens.'):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    for p in model.parameters():
        p.grad = None
    groups = [dict(params=[p for p in gr['params'] if p.requires_grad], lr=gr['lr'], name=gr['name']) for gr in model.get_learnable_params()]
    opt = FusedAdamW([gr for gr in groups if gr['params']], weight_decay=1e-4)
    drv = StepDriver(dp, opt, accum_steps=1, adaptive_clip=True, dp=dp)
    before = model.lora_arena.detach().clone()
    L2 = drv.step(inp['images'], inp['texts'], inp['modality_masks'], labels)
    st = opt.stats()
    assert st['non_finite'] == 0 and st['grad_norm'] > 0 and opt.step_count == 1
    assert float((model.lora_arena.detach() - before).abs().max()) > 0
    assert abs(float(L2['total_loss'].detach()) - float(L['total_loss'].detach())) <= 1e-3 * abs(float(L['total_loss'].detach()))


class _SpyAdamW:
    """FusedAdamW that keeps a copy of the gradients it is asked to apply."""

    def __init__(self, groups):
        from prcv2025reid_amd.trainer import FusedAdamW
        self.inner = FusedAdamW(groups, weight_decay=1e-4)
        self.seen = []

    def __getattr__(self, k):
        return getattr(self.inner, k)

    def step(self, **kw):
        self.seen.append([p.grad.detach().clone() for p in self.inner.params])
        return self.inner.step(**kw)


def test_config5_per_gpu_slice_full_size_accum2():
    """configs[4]: P=64,K=4 per GPU (B = 256, 1 024 images packed by the routing plan minus the masked ones), r=16, each non-RGB
    modality present with probability 0.7, accum_steps = 2 through StepDriver."""
    from prcv2025reid_amd.trainer import StepDriver
    C = 400
    model, cfg = _model(16, C)
    A, la, ba = _batch(model, 64, 4, 2000, 0.3, C)
    Bm, lb, bb = _batch(model, 64, 4, 2001, 0.3, C)
    for b in (ba, bb):
        frac = float(torch.stack([b['modality_mask'][m] for m in ('nir', 'sk', 'cp', 'text')]).mean())
        assert 0.6 < frac < 0.8 and float(b['modality_mask']['vis'].min()) == 1.0
    # the two micro-batches separately: head consistency, validity count, g_A and g_B
    grads = []
    for inp, labels, b in ((A, la, ba), (Bm, lb, bb)):
        out = model(**inp)
        assert out['bn_features'].shape == (256, 512)
        L = model.compute_loss(out, labels)
        _head_consistency(model, cfg, out, L, b['person_id'])
        assert int(L['ce_valid_cnt']) == 256                  # RGB is always present: every sample has a valid modality
        (L['total_loss'] / 2).backward()
        ps = [p for p in model.parameters() if p.requires_grad]
        grads.append([(p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for p in ps])
        for p in ps:
            assert p.grad is None or torch.isfinite(p.grad).all()
        # (a masked slot holds the modality's null token, but the fusion's key mask and masked mean and the SDM pairs all leave
        #  masked rows out -- model.py:141-149,586-622 -- so null tokens receive exactly zero gradient here, as in the reference)
        for p in model.parameters():
            p.grad = None
    groups = [dict(params=[p for p in gr['params'] if p.requires_grad], lr=gr['lr'], name=gr['name']) for gr in model.get_learnable_params()]
    spy = _SpyAdamW([gr for gr in groups if gr['params']])
    drv = StepDriver(model, spy, accum_steps=2, adaptive_clip=True)
    drv.step(A['images'], A['texts'], A['modality_masks'], la)
    assert len(spy.seen) == 0 and spy.inner.step_count == 0               # first micro-batch: accumulate only
    drv.step(Bm['images'], Bm['texts'], Bm['modality_masks'], lb)
    assert len(spy.seen) == 1 and spy.inner.step_count == 1
    order = {id(p): i for i, p in enumerate(p for p in model.parameters() if p.requires_grad)}
    worst = 0.0
    for p, g in zip(spy.inner.params, spy.seen[0]):
        want = grads[0][order[id(p)]] + grads[1][order[id(p)]]
        if float(want.abs().max()) == 0.0:
            assert float(g.abs().max()) == 0.0
            continue
        rel = float((g - want).norm() / want.norm())
        worst = max(worst, rel)
        assert rel <= 2e-3, rel          # same kernels on the same data; fp32 atomics of the dA/dB reductions reorder sums
    st = spy.inner.stats()
    assert st['non_finite'] == 0 and st['grad_norm'] > 0
    print(f'  accum 2: gradient seen by the optimizer vs (g_A + g_B) / 2: worst rel-L2 {worst:.2e}; |g| = {st["grad_norm"]:.3e}')
