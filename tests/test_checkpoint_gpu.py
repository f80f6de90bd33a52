"""GPU: a checkpoint written after a training step restores the model and the fused optimizer exactly."""
import pytest
import torch

from helpers import load_case, case_inputs

pytestmark = pytest.mark.gpu


def test_checkpoint_round_trip(tmp_path):
    from prcv2025reid_amd import checkpoint as ck
    from prcv2025reid_amd.trainer import FusedAdamW, StepDriver
    from test_model_gpu import build_model
    z, meta = load_case('tiny_train_frozen')
    cfg, arch, state, batch, tokens = case_inputs(meta)
    images = {m: t.cuda() for m, t in batch['images'].items()}
    masks = {m: t.cuda() for m, t in batch['modality_mask'].items()}
    labels = batch['person_id'].cuda()

    def groups(model):
        return [dict(params=[p for p in g['params'] if p.requires_grad], lr=g['lr'], name=g['name']) for g in model.get_learnable_params()]

    a = build_model(meta, state, True)
    opt = FusedAdamW(groups(a), weight_decay=1e-4)
    drv = StepDriver(a, opt)
    drv.step(images, batch['texts'], masks, labels)
    path = str(tmp_path / 'ckpt' / 'e1.pth')
    ck.save_checkpoint(a, opt, {'last_epoch': 1}, 1, 0.25, a.config, path)

    state2 = {k: v + 0.01 for k, v in state.items()}                       # a different starting point
    b = build_model(meta, state2, True)
    opt_b = FusedAdamW(groups(b), weight_decay=1e-4)
    info = ck.load_checkpoint(path, b, opt_b)
    assert info['epoch'] == 1 and info['best_map'] == 0.25 and info['scheduler_state_dict'] == {'last_epoch': 1}
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa) == set(sb)
    for k in sa:
        assert torch.equal(sa[k].cpu(), sb[k].cpu()), k
    assert opt_b.step_count == 1
    for x, y in zip(opt.exp_avg + opt.exp_avg_sq, opt_b.exp_avg + opt_b.exp_avg_sq):
        assert torch.equal(x, y)
    # dead keys survive the round trip: the file loads strictly into the reference's key set
    full = torch.load(path, map_location='cpu', weights_only=False)['model_state_dict']
    assert 'clip_encoder.clip_model.vision_model.post_layernorm.weight' in full and 'bn_neck.bn.num_batches_tracked' in full
    a.eval(); b.eval()
    with torch.no_grad():
        oa = a(images=images, texts=batch['texts'], modality_masks=masks)
        ob = b(images=images, texts=batch['texts'], modality_masks=masks)
    assert torch.equal(oa['bn_features'], ob['bn_features'])
