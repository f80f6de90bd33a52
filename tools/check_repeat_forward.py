"""Several models in one process, four identical no-grad forwards each: prints which outputs differ from the first (must be none).
The check that found the r03 forward-only race (adapter-gradient side products queued by a no-grad forward); tests/test_model_gpu.py
test_no_grad_forward_saves_nothing_and_is_reproducible is its pytest form.  usage: python tools/check_repeat_forward.py bf16,f16,bf16"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from prcv2025reid_amd.synthetic import synthetic_batch
from prcv2025reid_amd.weights import seeded_state

order = sys.argv[1].split(',')
C, P, K = 400, 16, 4
state = batch = tok = None
for fl in order:
    model = bench.build_model(0, 8, fl, C, regularisers=False)
    if state is None:
        state = seeded_state(model.arch, C, 0)
        batch = synthetic_batch(P, K, model.arch, seed=1000, num_classes=C)
        tok = model.tokenizer(batch['texts'], return_tensors='pt', padding=True, truncation=True, max_length=77)
    model.load_state_dict(state)
    labels = batch['person_id'].to('cuda:0')
    outs = []
    for rep in range(4):
        with torch.no_grad():
            out = model(images={m: t.to('cuda:0') for m, t in batch['images'].items()},
                        texts={k: v.to('cuda:0') for k, v in tok.items()}, modality_masks=batch['modality_mask'])
            torch.cuda.synchronize()
            if os.environ.get('WITH_LOSS', '1') == '1':
                L = model.compute_loss(out, labels)
                torch.cuda.synchronize()
        outs.append(out)
    torch.cuda.synchronize()
    for rep in range(1, 4):
        msgs = []
        for k in ('features', 'bn_features', 'logits'):
            d = (outs[rep][k] - outs[0][k]).abs()
            if float(d.max()) > 0:
                rows = (d.max(dim=1).values > 0).nonzero().flatten().tolist()
                msgs.append(f'{k}: max {float(d.max()):.3e} rows {rows[:8]}{"..." if len(rows) > 8 else ""} ({len(rows)})')
        for g in ('raw_modality_features', 'modality_features'):
            for m in outs[0][g]:
                d = (outs[rep][g][m] - outs[0][g][m]).abs()
                if float(d.max()) > 0:
                    rows = (d.max(dim=1).values > 0).nonzero().flatten().tolist()
                    msgs.append(f'{g}/{m}: max {float(d.max()):.3e} rows {rows[:8]} ({len(rows)})')
        for m in outs[0]['feature_masks']:
            if not torch.equal(outs[rep]['feature_masks'][m], outs[0]['feature_masks'][m]):
                msgs.append(f'mask {m} differs')
        print(fl, 'rep', rep, 'vs 0:', 'identical' if not msgs else '', flush=True)
        for s in msgs:
            print('     ', s, flush=True)
    del model, outs, out
    torch.cuda.empty_cache()
