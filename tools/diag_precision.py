"""Diagnostic (GPU box): where the bf16 error of the HIP path comes from, stage by stage, vs the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from helpers import load_case, case_inputs, case_config
from oracle import reid_oracle as O
from prcv2025reid_amd.model import CLIPBasedMultiModalReIDModel, apply_reference_freeze

name = sys.argv[1] if len(sys.argv) > 1 else 'full_eval_r8'
z, meta = load_case(name)
cfg, arch, state, batch, tokens = case_inputs(meta)
training = bool(meta['training'])
model = CLIPBasedMultiModalReIDModel(case_config(meta, device='cuda'))
model.set_num_classes(int(meta['num_classes'])); model.load_state_dict(state); apply_reference_freeze(model)
model.train(training); model.set_epoch(2)
images = {m: t.cuda() for m, t in batch['images'].items()}
masks = {m: t.cuda() for m, t in batch['modality_mask'].items()}
with torch.no_grad():
    out = model(images=images, texts=batch['texts'], modality_masks=masks)
def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).norm() / b.norm())
def unit_max(a, b):
    a = torch.nn.functional.normalize(a.double().cpu(), dim=1); b = torch.nn.functional.normalize(b.double().cpu(), dim=1)
    return float((a - b).abs().max())
print('case', name, 'training', training)
for m in out['raw_modality_features']:
    ref = torch.as_tensor(z[f'raw.{m}'])
    got = out['raw_modality_features'][m]
    cen = ref - ref.mean(0, keepdim=True)
    print(f'raw.{m}: relL2={rel(got, ref):.2e} unit max|d|={unit_max(got, ref):.2e}  (sample-varying part / total = {float(cen.norm()/ref.norm()):.3f})')
print('features relL2', rel(out['features'], torch.as_tensor(z['features'])), 'unit max', unit_max(out['features'], torch.as_tensor(z['features'])))
print('bn_features/8 max|d|', float((out['bn_features'].cpu()/8 - torch.as_tensor(z['bn_features'])/8).abs().max()))
# head amplification: oracle head on OUR raw features vs on reference raw features
raw_ours = {m: t.detach().cpu() for m, t in out['raw_modality_features'].items()}
sem = {m: (O.sdm_module(f, state) if training else f) for m, f in raw_ours.items()}
fm = {m: torch.as_tensor(z[f'fmask.{m}']) for m in raw_ours}
fused = O.feature_fusion(list(sem.values()), [fm[m] for m in sem], state, arch['fusion_num_heads'])
f, logits, _, _ = O.bn_neck(fused, state, training)
print('oracle-head(our raw) vs ref bn/8 max|d|', float((f/8 - torch.as_tensor(z['bn_features'])/8).abs().max()))
print('our head vs oracle-head(our raw) bn/8 max|d|', float((out['bn_features'].cpu()/8 - f/8).abs().max()))
