"""Experiment: does running two half-batches on two HIP streams overlap one's HBM-bound phases with the other's MFMA phases?
Compares (a) one model, P=16,K=4 on one stream with (b) two models, P=8,K=4 each, on two streams (same total work)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd.config import TrainingConfig
from prcv2025reid_amd.model import CLIPBasedMultiModalReIDModel, apply_reference_freeze
from prcv2025reid_amd.synthetic import synthetic_batch

dev = torch.device('cuda:0')

def make(P, K, seed):
    cfg = TrainingConfig(device='cuda:0', mer_lora_rank=8, contrastive_weight=0.1, seed=0)
    m = CLIPBasedMultiModalReIDModel(cfg); m.set_num_classes(400); apply_reference_freeze(m); m.set_epoch(2); m.train()
    b = synthetic_batch(P, K, m.arch, seed=seed, mask_drop=0.0, num_classes=400)
    images = {k: t.to(dev) for k, t in b['images'].items()}
    tok = m.tokenizer(b['texts'], return_tensors='pt', padding=True, truncation=True, max_length=77)
    tok = {k: v.to(dev) for k, v in tok.items()}
    return m, images, tok, b['modality_mask'], b['person_id'].to(dev)

def step(m, images, tok, masks, labels):
    out = m(images=images, texts=tok, modality_masks=masks)
    L = m.compute_loss(out, labels)
    L['total_loss'].backward()
    m.lora_arena.grad = None

full = make(16, 4, 1)
ha, hb = make(8, 4, 2), make(8, 4, 3)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(2):
    step(*full); step(*ha); step(*hb)
torch.cuda.synchronize()
def t_full(n=8):
    t0 = time.perf_counter()
    for _ in range(n): step(*full)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def t_seq(n=8):
    t0 = time.perf_counter()
    for _ in range(n): step(*ha); step(*hb)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def t_par(n=8):
    import threading
    def run(s, args):
        with torch.cuda.stream(s):
            for _ in range(n): step(*args)
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(sa, ha)), threading.Thread(target=run, args=(sb, hb))]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(2):
    print(f'full batch, one stream      : {t_full():.2f} ms')
    print(f'two halves, sequential      : {t_seq():.2f} ms')
    print(f'two halves, two streams     : {t_par():.2f} ms', flush=True)
