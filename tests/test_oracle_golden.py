"""CPU: the oracle (oracle/reid_oracle.py) against fixtures produced by the reference itself."""
import numpy as np
import pytest
import torch

from helpers import load_case, case_inputs, check_fingerprint, maxdiff, edge_inputs, MODDROP_CASES
from oracle import reid_oracle as O


def _run_oracle(name, need_grad, variant=None, keep=None):
    z, meta = load_case(name)
    cfg, arch, state, batch, tokens = case_inputs(meta)
    check_fingerprint(z, state)
    training = bool(meta['training'])
    if need_grad:
        for k, t in state.items():
            if t.dtype.is_floating_point and 'running_' not in k:
                if not meta['freeze'] or ('loras' in k or 'bn_neck' in k or 'null_tokens' in k):
                    t.requires_grad_(True)
    images, texts, masks = edge_inputs(batch, variant)
    if texts is not None and variant is not None:
        from prcv2025reid_amd.tokenizer import HashTokenizer
        tok = HashTokenizer(arch['text_vocab'], arch['text_bos_id'], arch['text_eos_id'], arch['text_max_len'])
        tokens = tok(texts, padding=True, truncation=True, max_length=77)
    out = O.forward(state, arch, images, tokens if texts is not None else None, masks, training, moddrop_keep=keep)
    return z, meta, state, batch, out


def _check_outputs(z, out, tol):
    if 'fused_modalities' in z.files:        # which modalities reached the fusion block / the loss (dropout removes some)
        assert list(out['modality_features'].keys()) == [str(x) for x in z['fused_modalities']]
        assert sorted(out['feature_masks'].keys()) == sorted(k[6:] for k in z.files if k.startswith('fmask.'))
    for k in ('features', 'bn_features', 'logits'):
        assert maxdiff(out[k].detach(), z[k]) < tol, k
    for m, t in out['raw_modality_features'].items():
        assert maxdiff(t.detach(), z[f'raw.{m}']) < tol, f'raw.{m}'
    for m, t in out['modality_features'].items():
        assert maxdiff(t.detach(), z[f'sem.{m}']) < tol, f'sem.{m}'
    for m, t in out['feature_masks'].items():
        assert maxdiff(t, z[f'fmask.{m}']) == 0


def _check_train(z, meta, state, batch, out, tol, gtol):
    L = O.compute_loss(out, batch['person_id'], ce_weight=meta['ce_weight'],
                       contrastive_weight=meta['contrastive_weight'], tau=meta['tau'])
    for k in ('total_loss', 'ce_loss', 'sdm_loss'):
        assert abs(float(L[k]) - float(z[k])) < tol, (k, float(L[k]), float(z[k]))
    assert L['ce_valid_cnt'] == int(z['ce_valid_cnt'])
    L['total_loss'].backward()
    sumsq = 0.0
    for k, t in state.items():
        if t.grad is not None:
            sumsq += float(t.grad.double().pow(2).sum())
    want = float(z['grad_sumsq'])
    assert abs(sumsq - want) <= 1e-3 * want, (sumsq, want)
    n = 0
    for f in z.files:
        if f.startswith('grad.'):
            g = state[f[5:]].grad
            assert g is not None, f
            ref = z[f]
            scale = max(1e-12, float(np.abs(ref).max()))
            assert maxdiff(g, ref) <= gtol * scale + 1e-9, f
            n += 1
    assert n > 10 or not meta.get('many_grads', 1)


@pytest.mark.parametrize('variant', ['nomask', 'single', 'textonly', 'deadrow'])
def test_oracle_edge_cases(variant):
    """Reference quirks of forward(): no masks => no image is encoded (model.py:367,386-389); one modality => identity
    fusion (:125-126,479-480); a sample without any valid modality => global mean in slot 0 (:141-149) and no CE term;
    text without a mask => all valid (:417-418)."""
    z, meta, state, batch, out = _run_oracle(f'tiny_edge_{variant}', True, variant=variant)
    _check_outputs(z, out, 5e-5)        # (B = 6 batch-statistics BN magnifies fp32 summation-order noise: 2.1e-5 seen on 'single')
    meta['many_grads'] = 0
    _check_train(z, meta, state, batch, out, 2e-5, 2e-4)


def test_oracle_edge_single_eval():
    z, meta, state, batch, out = _run_oracle('tiny_edge_single_eval', False, variant='single')
    _check_outputs(z, out, 2e-5)
    assert maxdiff(out['features'], out['raw_modality_features']['vis']) == 0.0       # identity: no fusion, no SDM module in eval


@pytest.mark.parametrize('name', sorted(MODDROP_CASES))
def test_oracle_modality_dropout_fixed_draws(name):
    """Batch-level modality dropout (model.py:434-474) with the reference's draws fixed by the fixture generator."""
    forced, _, variant = MODDROP_CASES[name]
    keep = dict(zip(('nir', 'sk', 'cp', 'text'), [v > 0.5 for v in forced]))
    z, meta, state, batch, out = _run_oracle(name, True, variant=variant, keep=keep)
    assert int(z['forced_used']) == 4
    _check_outputs(z, out, 5e-5)
    meta['many_grads'] = 0
    _check_train(z, meta, state, batch, out, 2e-5, 2e-4)


@pytest.mark.parametrize('name', ['tiny_train_frozen', 'tiny_train_all', 'tiny_train_r16_masked'])
def test_oracle_tiny_train(name):
    z, meta, state, batch, out = _run_oracle(name, True)
    _check_outputs(z, out, 2e-5)
    _check_train(z, meta, state, batch, out, 2e-5, 2e-4)
    # BN running-stat update (torch rule) -- used by the product's BN-neck too
    mu, var = out['bn_batch_mean'].detach(), out['bn_batch_var'].detach()
    B = out['features'].shape[0]
    rm = 0.9 * state['bn_neck.bn.running_mean'].detach() + 0.1 * mu
    rv = 0.9 * state['bn_neck.bn.running_var'].detach() + 0.1 * var * B / (B - 1)
    assert maxdiff(rm, z['bn_running_mean']) < 1e-5
    assert maxdiff(rv, z['bn_running_var']) < 1e-5


def test_oracle_tiny_eval():
    z, meta, state, batch, out = _run_oracle('tiny_eval', False)
    _check_outputs(z, out, 2e-5)
    n = torch.as_tensor(z['bn_features']).norm(dim=1)
    assert float((n - 8.0).abs().max()) < 1e-4           # SURVEY 8c known answer (4)


@pytest.mark.parametrize('name', ['full_p4k2_r4', 'full_p4k2_r8_masked', 'full_p4k2_r16_masked'])
def test_oracle_full_train(name):
    torch.set_num_threads(8)
    z, meta, state, batch, out = _run_oracle(name, True)
    _check_outputs(z, out, 1e-4)
    meta['many_grads'] = int(name != 'full_p4k2_r16_masked')
    _check_train(z, meta, state, batch, out, 1e-4, 1e-3)


def test_oracle_full_eval():
    z, meta, state, batch, out = _run_oracle('full_eval_r8', False)
    _check_outputs(z, out, 1e-4)


def test_sdm_known_answers():
    z = np.load(__import__('os').path.join(__import__('helpers').GOLDEN, 'sdm_known.npz'))
    assert float(z['quick_check']) == 3.3439764976501465      # models/sdm_loss.py:153-167, SURVEY section 4
    torch.manual_seed(0)
    qry = torch.randn(16, 768); gal = torch.randn(48, 768)
    ql = torch.randint(0, 10, (16,)); gl = torch.randint(0, 10, (48,))
    y = (ql.view(-1, 1) == gl.view(1, -1)).float()
    assert abs(float(O.sdm_loss(qry, gal, y, tau=0.2)) - 3.3439764976501465) < 2e-6
    y2 = y.clone(); y2[:5] = 0
    assert abs(float(O.sdm_loss(qry, gal, y2, tau=0.05)) - float(z['tau005_rows5zero'])) < 2e-6
    assert float(O.sdm_loss(qry, gal, torch.zeros_like(y))) == 0.0


def test_retrieval_metrics_golden():
    import os
    from helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'retrieval_metrics.npz'))
    Q, G = torch.as_tensor(z['Q']), torch.as_tensor(z['G'])
    qp, gp = torch.as_tensor(z['q_pid']), torch.as_tensor(z['g_pid'])
    mAP, top1 = O.reid_map(Q @ G.t(), qp, gp)
    assert abs(mAP - float(z['reid_map'])) < 1e-6 and abs(top1 - float(z['reid_top1'])) < 1e-9
    g_img = [f'g{i}' for i in range(G.shape[0])]
    q_img = [set(s.split('|')) - {''} for s in z['q_img'].tolist()]
    r = O.rank_and_metrics(Q, qp, G, gp, q_img, g_img)
    assert r['num_queries'] == int(z['rm_n'])
    for k, kk in (('mAP', 'rm_mAP'), ('R@1', 'rm_r1'), ('R@5', 'rm_r5'), ('R@10', 'rm_r10')):
        assert abs(r[k] - float(z[kk])) < 1e-6, k
    r = O.rank_and_metrics(Q, qp, G, gp, None, None)
    assert r['num_queries'] == int(z['rn_n'])
    for k, kk in (('mAP', 'rn_mAP'), ('R@1', 'rn_r1'), ('R@5', 'rn_r5'), ('R@10', 'rn_r10')):
        assert abs(r[k] - float(z[kk])) < 1e-6, k
