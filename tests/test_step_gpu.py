"""GPU: fused optimizer step (csrc/optim.hip through the C ABI) against the CPU step oracle, and the step driver
against the same sequence made of torch library calls (what train.py:975-1047 runs)."""
import numpy as np
import pytest
import torch

from helpers import load_case, case_inputs
from oracle import step_oracle as so

pytestmark = pytest.mark.gpu


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g) for s in shapes], g


@pytest.mark.parametrize('adaptive', [True, False])
def test_fused_adamw_vs_oracle(adaptive):
    from prcv2025reid_amd.trainer import FusedAdamW
    shapes = [(70001,), (512, 33), (5,), (64, 16), (3,)]            # odd sizes: scalar tails of the 16-byte loops
    params, g = _params(3, shapes)
    dev_p = [torch.nn.Parameter(p.clone().cuda()) for p in params]
    groups = [dict(params=[dev_p[0], dev_p[1]], lr=5e-5, name='mer_loras'),
              dict(params=[dev_p[2], dev_p[4]], lr=3e-3, name='classification_head'),
              dict(params=[dev_p[3]], lr=5e-5, name='other_modules', weight_decay=0.0)]
    opt = FusedAdamW(groups, weight_decay=1e-4)
    orc = so.StepOracle([p.clone() for p in params], [([0, 1], 5e-5, 1e-4), ([2, 4], 3e-3, 1e-4), ([3], 5e-5, 0.0)])
    order = [0, 1, 2, 4, 3]                                          # table order = group order
    for step in range(14):
        scale = float(10.0 ** ((step % 5) - 3))
        grads = [torch.randn(p.shape, generator=g) * scale for p in params]
        if step in (2, 12):
            grads[0][17] = float('nan'); grads[1][3, 4] = float('-inf'); grads[4][2] = float('inf')
        for p, gr in zip(dev_p, grads):
            p.grad.copy_(gr)
        opt.step(adaptive_clip=adaptive, record_norm=True, fixed_max_norm=0.5, zero_grad=True)
        want = orc.step([gr.clone() for gr in grads], adaptive=adaptive, record=True, fixed_max_norm=0.5)
        got = opt.stats()
        assert got['non_finite'] == want['non_finite']
        assert abs(got['grad_norm'] - want['grad_norm']) <= 2e-6 * want['grad_norm']
        assert abs(got['max_norm'] - want['max_norm']) <= 1e-6 * want['max_norm'], (step, got, want)
        assert abs(got['clip_coef'] - want['clip_coef']) <= 2e-6
        for p, o in zip(dev_p, orc.params):
            assert torch.allclose(p.detach().cpu(), o, rtol=5e-6, atol=1e-7), step
            assert float(p.grad.abs().max()) == 0.0                  # cleared by the step kernel
    assert orc.history and len(orc.history) == 14 if adaptive else True
    del order


def test_step_driver_tiny_model():
    """Two accumulation windows through StepDriver == the same windows through torch calls on a twin model."""
    from prcv2025reid_amd.model import CLIPBasedMultiModalReIDModel, apply_reference_freeze
    from prcv2025reid_amd.trainer import FusedAdamW, StepDriver, warmup_cosine_lambda
    from test_model_gpu import build_model
    z, meta = load_case('tiny_train_frozen')
    cfg, arch, state, batch, tokens = case_inputs(meta)
    images = {m: t.cuda() for m, t in batch['images'].items()}
    masks = {m: t.cuda() for m, t in batch['modality_mask'].items()}
    labels = batch['person_id'].cuda()
    lm = warmup_cosine_lambda(10, 3)

    def trainables(model):
        gs = model.get_learnable_params()
        return [dict(params=[p for p in g['params'] if p.requires_grad], lr=g['lr'], name=g['name']) for g in gs]

    a = build_model(meta, state, True)
    opt = FusedAdamW(trainables(a), weight_decay=1e-4)
    drv = StepDriver(a, opt, accum_steps=2, adaptive_clip=True)
    drv.start_epoch(2, lm)
    for _ in range(4):
        La = drv.step(images, batch['texts'], masks, labels)

    b = build_model(meta, state, True)
    init = {k: p.detach().clone() for k, p in b.named_parameters() if p.requires_grad}
    groups = [g for g in trainables(b) if g['params']]
    topt = torch.optim.AdamW(groups, weight_decay=1e-4, foreach=False)
    for g in topt.param_groups:
        g['lr'] = g['lr'] * lm(1)
    hist = []
    for bi in range(4):
        if bi % 2 == 0:
            topt.zero_grad(set_to_none=True)
        out = b(images=images, texts=batch['texts'], modality_masks=masks)
        Lb = b.compute_loss(out, labels)
        (Lb['total_loss'] / 2).backward()
        if (bi + 1) % 2 == 0:
            ps = [p for g in groups for p in g['params']]
            gr = [p.grad for p in ps if p.grad is not None]
            so.sanitize_grads(gr)
            n = so.total_norm([x.cpu() for x in gr])
            if bi % 200 == 0:
                hist.append(n)
            torch.nn.utils.clip_grad_norm_(ps, max_norm=so.adaptive_max_norm(hist))
            topt.step()
    La = {k: v.detach() if torch.is_tensor(v) else v for k, v in La.items()}; Lb = {k: v.detach() if torch.is_tensor(v) else v for k, v in Lb.items()}
    assert abs(float(La['total_loss']) - float(Lb['total_loss'])) <= 2e-3 * max(1.0, abs(float(Lb['total_loss'])))
    pa = dict(a.named_parameters()); pb = dict(b.named_parameters())
    moved = 0
    for k, p in pa.items():
        if not p.requires_grad:
            continue
        q = pb[k]
        delta_a = float((p.detach() - q.detach()).float().norm())
        step_sz = float((q.detach() - init[k]).float().norm())
        # the twin differs only through fp32-atomic summation order inside the dA/dB GEMMs: Adam's sign-like first steps
        # turn a flipped near-zero gradient entry into a 2*lr difference, so compare in the norm of the step taken
        assert delta_a <= 0.05 * step_sz + 1e-7, (k, delta_a, step_sz)
        moved += step_sz > 1e-9
    assert moved >= 3


def test_graphed_step_matches_eager():
    """HIP-graph replay of the step == the eager StepDriver on a twin model (same kernels, same order)."""
    from prcv2025reid_amd.trainer import FusedAdamW, StepDriver, GraphedStep
    from test_model_gpu import build_model
    z, meta = load_case('tiny_train_frozen')
    cfg, arch, state, batch, tokens = case_inputs(meta)
    images = {m: t.cuda() for m, t in batch['images'].items()}
    masks = dict(batch['modality_mask'])
    labels = batch['person_id'].cuda()

    def make():
        m = build_model(meta, state, True)
        gs = [dict(params=[p for p in g['params'] if p.requires_grad], lr=g['lr'], name=g['name']) for g in m.get_learnable_params()]
        return m, StepDriver(m, FusedAdamW(gs, weight_decay=1e-4))

    a, da = make()
    tok = a.tokenizer(batch['texts'], return_tensors='pt', padding=True, truncation=True, max_length=77)
    tok = {k: v.cuda() for k, v in tok.items()}
    init = {k: p.detach().clone() for k, p in a.named_parameters() if p.requires_grad}
    g = GraphedStep(da, images, tok, masks, labels, warmup=2)          # 2 eager steps + the capture pass (not executed)
    for _ in range(3):
        La = g.step(images, tok, masks, labels)
    torch.cuda.synchronize()
    b, db = make()
    for _ in range(5):
        Lb = db.step(images, tok, masks, labels)
    assert da.opt.step_count == db.opt.step_count == 5
    sa, sb = da.opt.stats(), db.opt.stats()
    assert abs(sa['grad_norm'] - sb['grad_norm']) <= 2e-2 * sb['grad_norm'], (sa, sb)
    assert abs(float(La['total_loss']) - float(Lb['total_loss'])) <= 2e-3 * max(1.0, abs(float(Lb['total_loss'])))
    pb = dict(b.named_parameters())
    for k, p in a.named_parameters():
        if p.requires_grad:
            step_sz = float((pb[k].detach() - init[k]).float().norm())
            assert float((p.detach() - pb[k].detach()).float().norm()) <= 0.05 * step_sz + 1e-7, k
    with pytest.raises(ValueError):
        bad = dict(masks); bad['nir'] = torch.zeros_like(masks['nir'])
        g.step(images, tok, bad, labels)


def test_graphed_step_with_training_regularisers():
    """DropPath / dropout masks come from this package's device generators: registered with the graph, every replay draws
    fresh ones (two replays on identical inputs and weights-frozen-in-place give different losses); active modality dropout
    (a host-side draw) is refused."""
    from prcv2025reid_amd.trainer import FusedAdamW, StepDriver, GraphedStep
    from test_model_gpu import build_model
    z, meta = load_case('tiny_train_frozen')
    cfg, arch, state, batch, tokens = case_inputs(meta)
    images = {m: t.cuda() for m, t in batch['images'].items()}
    masks = dict(batch['modality_mask'])
    labels = batch['person_id'].cuda()
    m = build_model(meta, state, True)
    m.drop_path = 0.3; m.dropout_rate = 0.5; m.fusion_dropout = 0.1; m.sdm_dropout = 0.1
    m.seed_stochastic(5)
    gs = [dict(params=[p for p in g['params'] if p.requires_grad], lr=0.0, name=g['name']) for g in m.get_learnable_params()]
    drv = StepDriver(m, FusedAdamW(gs, weight_decay=0.0))              # lr 0: the weights stay put, only the masks change
    tok = m.tokenizer(batch['texts'], return_tensors='pt', padding=True, truncation=True, max_length=77)
    tok = {k: v.cuda() for k, v in tok.items()}
    g = GraphedStep(drv, images, tok, masks, labels, warmup=1)
    losses = []
    for _ in range(3):
        L = g.step(images, tok, masks, labels)
        losses.append(float(L['total_loss']))
    assert all(l == l and abs(l) < 1e4 for l in losses)
    assert len(set(losses)) == 3, losses                                # fresh masks on every replay
    m.config.modality_dropout = 0.15; m.config.modality_dropout_warmup_epochs = 0; m.set_epoch(2)
    with pytest.raises(ValueError):
        GraphedStep(drv, images, tok, masks, labels, warmup=1)
