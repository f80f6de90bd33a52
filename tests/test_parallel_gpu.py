"""GPU: the PRODUCT data-parallel path -- two ranks (gloo, both on the one GPU of the test box) through
``parallel.DataParallel`` on the HIP model: the all-gathered global head (``DataParallel._gather_fn`` ->
``model.forward(gather_fn=...)``), the rank-local slice of the feature gradient, ``reduce_grads`` (LoRA arena + null tokens
summed, head gradients averaged) and the fused optimizer.

Checked against ONE process evaluating the global batch: same loss on every rank, summed LoRA-arena gradient equal to the
single-process gradient, and -- after three optimizer steps -- parameters bit-identical across the ranks.
(RCCL itself needs one GPU per rank: the driver's N = 2, 4, 8 bench runs exercise it; the arithmetic of the path is this test.)
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

P, K, C, WSEED, DSEED = 4, 2, 6, 4, 31


def _tiny_cfg():
    from prcv2025reid_amd.config import TrainingConfig
    return TrainingConfig(device='cuda:0', mer_lora_rank=4, contrastive_weight=0.1, vision_hidden_dim=128, vision_layers=2,
                          vision_heads=2, vision_mlp_dim=256, text_layers=2, text_mlp_dim=1024, text_vocab=1024,
                          text_eos_id=1023, text_bos_id=1022, seed=5, compute_dtype='f16', init='seeded',
                          drop_path=0.0, modality_dropout=0.0, dropout_rate=0.0, fusion_dropout=0.0, sdm_dropout=0.0)


def _build():
    from prcv2025reid_amd.config import arch_of
    from prcv2025reid_amd.model import CLIPBasedMultiModalReIDModel, apply_reference_freeze
    from prcv2025reid_amd.weights import seeded_state
    cfg = _tiny_cfg()
    model = CLIPBasedMultiModalReIDModel(cfg)
    model.set_num_classes(C)
    model.load_state_dict(seeded_state(arch_of(cfg), C, WSEED))
    apply_reference_freeze(model)
    model.set_epoch(2); model.train()
    return model


def _batch(model):
    from prcv2025reid_amd.synthetic import synthetic_batch
    b = synthetic_batch(P, K, model.arch, seed=DSEED, mask_drop=0.3, num_classes=C)
    tok = model.tokenizer(b['texts'], return_tensors='pt', padding=True, truncation=True, max_length=77)
    return b, tok


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from prcv2025reid_amd.parallel import DataParallel
    from prcv2025reid_amd.trainer import FusedAdamW, StepDriver
    model = _build()
    b, tok = _batch(model)
    n = P * K // world
    sl = slice(rank * n, (rank + 1) * n)                                   # whole identities per rank (K = 2 rows each)
    images = {m: t[sl].cuda() for m, t in b['images'].items()}
    masks = {m: t[sl] for m, t in b['modality_mask'].items()}
    tokens = {k: v[sl].cuda() for k, v in tok.items()}
    labels = b['person_id'][sl].cuda()
    dp = DataParallel(model)
    assert dp.world == world and dp.rank == rank
    out = dp.forward(images=images, texts=tokens, modality_masks=masks)
    assert out['bn_features'].shape[0] == P * K                            # the head saw the GLOBAL batch
    L = dp.compute_loss(out, labels)
    L['total_loss'].backward()
    dp.reduce_grads()
    named = dict(model.named_parameters())
    torch.save({'loss': float(L['total_loss'].detach()), 'sdm': float(L['sdm_loss'].detach()), 'lora': model.lora_arena.grad.cpu(),
                'null': {k: p.grad.cpu() for k, p in named.items() if k.startswith('null_tokens.') and p.grad is not None},
                'bnw': named['bn_neck.bn.weight'].grad.cpu(), 'cls': named['bn_neck.classifier.weight'].grad.cpu()},
               os.path.join(tmp, f'r{rank}.pt'))
    # three optimizer steps through the step driver: replicas must stay bit-identical
    for p in model.parameters():
        p.grad = None
    groups = [dict(params=[p for p in g['params'] if p.requires_grad], lr=g['lr'], name=g['name']) for g in model.get_learnable_params()]
    opt = FusedAdamW([g for g in groups if g['params']], weight_decay=1e-4)
    drv = StepDriver(dp, opt, accum_steps=1, adaptive_clip=True, dp=dp)
    for _ in range(3):
        drv.step(images, tokens, masks, labels)
    spread = dp.params_in_sync()
    # gradient accumulation through the data-parallel driver (train.py:833-834,895: accum 2): the all-reduce and the optimizer run on
    # every second micro-batch only; masks differ from rank to rank (mask_drop 0.3) and between the two micro-batches (rows reversed)
    drv2 = StepDriver(dp, opt, accum_steps=2, adaptive_clip=True, dp=dp)
    rev = torch.arange(n - 1, -1, -1)
    images_b = {m: t[rev.to(t.device)] for m, t in images.items()}
    masks_b = {m: t[rev] for m, t in masks.items()}
    tokens_b = {k: v[rev.to(v.device)] for k, v in tokens.items()}
    labels_b = labels[rev.to(labels.device)]
    count0 = opt.step_count
    for i in range(4):
        if i % 2 == 0:
            drv2.step(images, tokens, masks, labels)
        else:
            drv2.step(images_b, tokens_b, masks_b, labels_b)
    spread2 = dp.params_in_sync()
    torch.save({'spread': spread, 'spread_accum2': spread2, 'opt_steps_accum2': opt.step_count - count0,
                'bucket_views': all(p.grad is not None and p.grad.data_ptr() == ptr for p, ptr in zip(dp._bucket['params'], dp._bucket['ptrs']))},
               os.path.join(tmp, f's{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_global_batch(tmp_path):
    world = 2
    port = 29700 + (os.getpid() % 1500)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    model = _build()
    b, tok = _batch(model)
    out = model(images={m: t.cuda() for m, t in b['images'].items()}, texts={k: v.cuda() for k, v in tok.items()},
                modality_masks=b['modality_mask'])
    L = model.compute_loss(out, b['person_id'].cuda())
    L['total_loss'].backward()
    ref_loss = float(L['total_loss'].detach())
    ref_lora = model.lora_arena.grad.cpu()
    named = dict(model.named_parameters())
    outs = [torch.load(os.path.join(str(tmp_path), f'r{r}.pt')) for r in range(world)]
    for r, o in enumerate(outs):
        assert abs(o['loss'] - ref_loss) <= 2e-6 * max(1.0, abs(ref_loss)), (r, o['loss'], ref_loss)
        rel = float((o['lora'] - ref_lora).norm() / ref_lora.norm())
        # packed batches of 4 vs 8 samples change tile shapes / summation order of the f16 kernels, not the function
        assert rel < 2e-3, (r, rel)
        for k, g in o['null'].items():
            gr = named[k].grad.cpu() if named[k].grad is not None else torch.zeros_like(g)     # (an unused null token has no gradient)
            assert float((g - gr).abs().max()) <= 1e-5 + 2e-3 * float(gr.abs().max()), k
        for key, pk in (('bnw', 'bn_neck.bn.weight'), ('cls', 'bn_neck.classifier.weight')):
            gr = named[pk].grad.cpu()
            assert float((o[key] - gr).norm() / gr.norm().clamp_min(1e-20)) < 2e-3, key
    assert torch.equal(outs[0]['lora'], outs[1]['lora'])                    # all-reduced: identical bits on both ranks
    assert abs(outs[0]['loss'] - outs[1]['loss']) <= 1e-6
    for r in range(world):
        sr = torch.load(os.path.join(str(tmp_path), f's{r}.pt'))
        assert sr['spread'] == 0.0
        assert sr['spread_accum2'] == 0.0 and sr['opt_steps_accum2'] == 2      # four micro-batches = two optimizer steps, replicas identical
        assert sr['bucket_views']                                              # small gradients live in ONE flat bucket (two collectives per step)
    print(f'  DP(2 ranks) loss {outs[0]["loss"]:.6f} vs single process {ref_loss:.6f}')


def test_bench_gpus2_spawn_and_relay_gloo():
    """``bench.py --gpus 2`` with no WORLD_SIZE: the CPU-only parent spawns the ranks, watches all of them and relays rank 0's line
    (gloo rehearsal: both ranks on the one GPU of the test box; RCCL needs one GPU per rank).  Checks the keys the driver's scaling
    run relies on."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--P', '2', '--K', '2', '--steps', '2',
           '--warmup', '1', '--no-cpu-baseline', '--no-retrieval', '--no-parity', '--no-second-flavor', '--no-kernel-events',
           '--spawn-timeout', '900']
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1000, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 8 and line['scaling'] == 'weak'
    assert line['params_spread_over_ranks'] == 0.0 and line['loss_spread_over_ranks'] <= 1e-5
    assert line['n1_same_workload']['value'] > 0 and line['scaling_vs_n1_same_workload'] > 0
    assert line['rccl_world'] == 0 and line['backend'] == 'gloo'        # (a rehearsal says so; nccl runs report rccl_world = N)
