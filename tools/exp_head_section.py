"""How long is the head section (end of the vision forward -> start of the vision backward) on the GPU, without a profiler?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd.config import TrainingConfig
from prcv2025reid_amd.model import CLIPBasedMultiModalReIDModel, apply_reference_freeze
from prcv2025reid_amd.parallel import DataParallel
from prcv2025reid_amd.synthetic import synthetic_batch
from prcv2025reid_amd.trainer import FusedAdamW, StepDriver

dev = torch.device('cuda', 0)
cfg = TrainingConfig(device='cuda:0', mer_lora_rank=8, contrastive_weight=0.1, seed=0)
model = CLIPBasedMultiModalReIDModel(cfg); model.set_num_classes(400); apply_reference_freeze(model); model.set_epoch(2); model.train()
dp = DataParallel(model)
batch = synthetic_batch(16, 4, model.arch, seed=1000, mask_drop=0.0, num_classes=400, label_offset=0)
images = {m: t.to(dev) for m, t in batch['images'].items()}
masks = batch['modality_mask']
tok = model.tokenizer(batch['texts'], return_tensors='pt', padding=True, truncation=True, max_length=77)
tokens = {k: v.to(dev) for k, v in tok.items()}
labels = batch['person_id'].to(dev)
groups = [dict(params=[p for p in g['params'] if p.requires_grad], lr=g['lr'], name=g['name']) for g in model.get_learnable_params()]
opt = FusedAdamW([g for g in groups if g['params']], weight_decay=1e-4)
driver = StepDriver(dp, opt, accum_steps=1, adaptive_clip=True, dp=dp)

ev = {}
orig = model._vision_apply
cpu_t = {}
def wrapped(*a, **k):
    out = orig(*a, **k)
    e = torch.cuda.Event(enable_timing=True); e.record(); ev['fwd_end'] = e; cpu_t['fwd_end'] = time.perf_counter()
    def hook(g):
        e2 = torch.cuda.Event(enable_timing=True); e2.record(); ev['bwd_start'] = e2; cpu_t['bwd_start'] = time.perf_counter()
        return g
    out.register_hook(hook)
    return out
model._vision_apply = wrapped
for _ in range(5):
    driver.step(images, tokens, masks, labels)
torch.cuda.synchronize()
res = []
for _ in range(10):
    s = torch.cuda.Event(enable_timing=True); s.record(); c0 = time.perf_counter()
    driver.step(images, tokens, masks, labels)
    c1 = time.perf_counter()
    t = torch.cuda.Event(enable_timing=True); t.record(); torch.cuda.synchronize()
    res.append((s.elapsed_time(ev['fwd_end']), ev['fwd_end'].elapsed_time(ev['bwd_start']), ev['bwd_start'].elapsed_time(t), s.elapsed_time(t),
                (cpu_t['fwd_end'] - c0) * 1e3, (cpu_t['bwd_start'] - c0) * 1e3, (c1 - c0) * 1e3))
for r in res:
    print('gpu: fwd %.2f head %.2f bwd+opt %.2f total %.2f ms | cpu: fwd_end at %.2f, bwd_start at %.2f, step returns at %.2f ms' % r)
