#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the training hot path over one synthetic P x K batch per GPU:
zero_grad -> forward (4 vision modalities + text through the HIP executor) -> CE + SDM losses ->
backward (hand-written HIP backward, LoRA/bn_neck/null-token gradients = the reference's default
trainable set, train.py:1418-1425) -> gradient all-reduce (N > 1) -> AdamW step.
Workload at every N: BASELINE.json configs[1] per GPU (P=16, K=4, LoRA r=8, masks all-on, ViT-B/16 + CLIP
text random-init, 400 identities), inputs resident in HBM; weak scaling.

One JSON line on stdout (rank 0).  Besides the contract keys it carries
  roofline      -- dominant kernel (mer_gemm_kernel<256,256,2,4> + <128,256,2,4>, bf16 MFMA): algorithmic FLOPs of every launch in
                   the timed region / its duration measured with HIP events on the launch stream
  cpu_baseline  -- the CPU oracle (oracle/reid_oracle.py, kind "port") timed on this host on a bounded sample
  retrieval     -- eval queries/s of the fused cosine top-10 on 10k x 200k x 512 (BASELINE.json configs[3])
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0          # HBM3E spec (6.3 TB/s measured float4 copy), same guide
FLOP_PER_INSTANCE = 292.0e9    # fwd 148.25 + bwd (frozen backbone) 143.76 GFLOP at r=8, T=77


def log(msg):
    print(f'[bench] {msg}', file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--P', type=int, default=16)
    ap.add_argument('--K', type=int, default=4)
    ap.add_argument('--rank', type=int, default=8, help='LoRA rank')
    ap.add_argument('--mask-drop', type=float, default=0.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-retrieval', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--optimizer', default='fused', choices=['fused', 'torch'])
    ap.add_argument('--graph', default='off', choices=['auto', 'on', 'off'],
                    help='replay the step as a HIP graph (single process only; measured ~1 ms/step SLOWER than eager launches on MI355X, r01)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL); gloo only for rehearsals')
    ap.add_argument('--compute-dtype', default=None, choices=[None, 'bf16', 'f16'])
    return ap.parse_args()


def pmc_traffic(*kernels):
    """Launch-weighted HBM bytes per launch of `kernels` from the committed PMC summary (collected with separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, corrected as the microarch guide prescribes)."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    n = sum(d[k]['launches'] for k in kernels if k in d)
    if n == 0:
        return None
    return sum(d[k]['traffic_bytes_per_launch'] * d[k]['launches'] for k in kernels if k in d) / n


def cpu_baseline():
    """Oracle (fp32 CPU restatement, validated against the reference) on BASELINE config 1: P=4,K=2, r=4, one
    SDM+CE step = forward + loss + backward, no optimizer.  1 warm-up + 3 timed."""
    from oracle import reid_oracle as O
    from prcv2025reid_amd.config import TrainingConfig, arch_of
    from prcv2025reid_amd.synthetic import synthetic_batch
    from prcv2025reid_amd.tokenizer import HashTokenizer
    from prcv2025reid_amd.weights import seeded_state
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))            # the GPU box gives one GPU a 16-core share
    torch.set_num_threads(cores)
    cfg = TrainingConfig(device='cpu', mer_lora_rank=4, contrastive_weight=0.1)
    arch = arch_of(cfg)
    state = seeded_state(arch, 16, 0)
    for k, t in state.items():
        if 'loras' in k or 'bn_neck' in k or 'null_tokens' in k:
            if t.dtype.is_floating_point and 'running_' not in k:
                t.requires_grad_(True)
    batch = synthetic_batch(4, 2, arch, seed=1, num_classes=16)
    tok = HashTokenizer()(batch['texts'])
    times = []
    for it in range(4):
        log(f'cpu_baseline: oracle step {it} on {cores} threads')
        t0 = time.perf_counter()
        out = O.forward(state, arch, batch['images'], tok, batch['modality_mask'], True)
        L = O.compute_loss(out, batch['person_id'], contrastive_weight=0.1, tau=0.2)
        L['total_loss'].backward()
        for t in state.values():
            t.grad = None
        if it > 0:
            times.append(time.perf_counter() - t0)
    t = sum(times) / len(times)
    return {'value': 8.0 / t, 'unit': 'instances/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'oracle fp32, P=4,K=2 (8 instances), LoRA r=4, fwd+loss+bwd, 1 warm-up + 3 timed steps',
            'seconds_per_step': t}


def retrieval_bench(dev):
    from prcv2025reid_amd.retrieval import GalleryIndex
    Nq, Ng, D, k = 10000, 200000, 512, 10
    g = torch.Generator(device=dev).manual_seed(2)
    Q = torch.nn.functional.normalize(torch.randn(Nq, D, device=dev, generator=g), dim=1)
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device=dev, generator=g), dim=1)
    index = GalleryIndex(G, normalized=True)           # gallery resident in HBM (fp32 + 16-bit copies)
    for _ in range(2):
        idx, sc = index.topk(Q, k=k, normalized=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        idx, sc = index.topk(Q, k=k, normalized=True)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    # exactness spot check on 64 queries against fp32 matmul + stable argsort
    ref = torch.argsort((Q[:64] @ G.t()), dim=1, descending=True, stable=True)[:, :k]
    exact = bool((ref == idx[:64].long()).all())
    out = {'queries_per_s': Nq / t, 'ms': t * 1e3, 'Nq': Nq, 'Ng': Ng, 'D': D, 'k': k, 'tflops': 2.0 * Nq * Ng * D / t / 1e12,
           'mfma_frac': 2.0 * Nq * Ng * D / t / 1e12 / PEAK_BF16_TFLOPS,
           'compulsory_bytes': (Nq + Ng) * D * 2 + Nq * k * 4, 'top10_equals_fp32_argsort_on_64_queries': exact}
    # the reference's one-query-at-a-time form (eval_mm_protocol.py:401-455): one pass over the fp32 gallery per call, HBM-bound
    q1 = Q[:1].contiguous()
    for _ in range(3):
        i1, s1 = index.topk(q1, k=k, normalized=True)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        i1, s1 = index.topk(q1, k=k, normalized=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    out['single_query'] = {'us_per_call': us, 'queries_per_s': 1e6 / us, 'gallery_bytes_fp32': Ng * D * 4,
                           'gallery_GBps_end_to_end': Ng * D * 4 / (us * 1e-6) / 1e9, 'peak_GBps': PEAK_HBM_GBS,
                           'hbm_frac_end_to_end': Ng * D * 4 / (us * 1e-6) / 1e9 / PEAK_HBM_GBS,
                           'equals_batched_top10': bool(torch.equal(i1, idx[:1]))}
    # full MM-protocol metrics (mAP over the whole ranking + CMC) of the same 10k x 200k problem, gallery pids randint(0, 1000)
    from prcv2025reid_amd.evaluate import ProtocolEvaluator
    gp = torch.randint(0, 1000, (Ng,), device=dev, generator=g)
    qp = torch.randint(0, 1000, (Nq,), device=dev, generator=g)
    ev = ProtocolEvaluator(G, gp, normalized=True)
    ev.rank_and_metrics(Q[:2048], qp[:2048])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = ev.rank_and_metrics(Q, qp)
    torch.cuda.synchronize()
    te = time.perf_counter() - t0
    out['protocol_eval'] = {'queries_per_s': Nq / te, 'ms': te * 1e3, 'mAP': m['mAP'], 'R@1': m['R@1'], 'num_queries': m['num_queries'],
                            'what': 'fp32-grade scores (split 16-bit MFMA GEMM) + AP/CMC over the full ranking, on device'}
    return out


def encode_bench(model, dev, arch):
    """Feature extraction of the evaluation protocol (tools/eval_mm_protocol.py:300-398): gallery = RGB images, queries =
    modality combinations; eval mode, no autograd.  Reported next to the retrieval numbers because end-to-end
    "eval queries/sec" is bounded by whichever of (encode, rank) is slower."""
    was_training = model.training
    model.eval()
    g = torch.Generator(device=dev).manual_seed(3)
    out = {}
    H = arch['image_size']
    cases = {'gallery_vis_b256': (['vis'], False, 256), 'query_quad_nir_sk_cp_text_b64': (['nir', 'sk', 'cp'], True, 64),
             'query_single_sk_b256': (['sk'], False, 256)}
    with torch.no_grad():
        for name, (mods, text, B) in cases.items():
            images = {m: torch.randn(B, 3, H, H, device=dev, generator=g) for m in mods}
            masks = {m: torch.ones(B) for m in mods}
            tokens = None
            if text:
                tok = model.tokenizer(['a person walking'] * B, return_tensors='pt', padding=True, truncation=True, max_length=77)
                tokens = {k: v.to(dev) for k, v in tok.items()}
                masks['text'] = torch.ones(B)
            for _ in range(2):
                f = model(images=images, texts=tokens, modality_masks=masks, return_features=True)
            torch.cuda.synchronize()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                f = model(images=images, texts=tokens, modality_masks=masks, return_features=True)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / reps
            out[name] = {'rows_per_s': B / t, 'ms': t * 1e3, 'batch': B}
    model.train(was_training)
    return out


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    local = local % max(1, ndev)                   # (rehearsal: several ranks may share one GPU under gloo)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)
    from prcv2025reid_amd import ops
    from prcv2025reid_amd.config import TrainingConfig
    from prcv2025reid_amd.model import CLIPBasedMultiModalReIDModel, apply_reference_freeze
    from prcv2025reid_amd.parallel import DataParallel
    from prcv2025reid_amd.synthetic import synthetic_batch

    C = 400
    cfg = TrainingConfig(device=f'cuda:{local}', mer_lora_rank=args.rank, contrastive_weight=0.1, seed=0,
                         compute_dtype=args.compute_dtype)
    model = CLIPBasedMultiModalReIDModel(cfg)
    model.set_num_classes(C)
    apply_reference_freeze(model)
    model.set_epoch(2)
    model.train()
    dp = DataParallel(model)
    P, K = args.P, args.K
    B = P * K
    batch = synthetic_batch(P, K, model.arch, seed=1000 + rank, mask_drop=args.mask_drop, num_classes=C, label_offset=rank * P)
    images = {m: t.to(dev) for m, t in batch['images'].items()}
    masks = batch['modality_mask']                      # host tensors, as the reference's collate produces them
    tok = model.tokenizer(batch['texts'], return_tensors='pt', padding=True, truncation=True, max_length=77)
    tokens = {k: v.to(dev) for k, v in tok.items()}     # pre-tokenised, resident in HBM
    labels = batch['person_id'].to(dev)
    groups = [dict(params=[p for p in g['params'] if p.requires_grad], lr=g['lr'], name=g['name'])
              for g in model.get_learnable_params()]
    groups = [g for g in groups if g['params']]
    graphed = False
    if args.optimizer == 'fused':
        # the reference's step (train.py:975-1047): sanitise, adaptive clip, AdamW -- three fused launches, no host sync
        from prcv2025reid_amd.trainer import FusedAdamW, StepDriver, GraphedStep
        opt = FusedAdamW(groups, weight_decay=1e-4)
        driver = StepDriver(dp, opt, accum_steps=1, adaptive_clip=True, dp=dp)
        gstep = None
        if world == 1 and args.graph != 'off':
            try:       # single process: the whole step replayed as one HIP graph (inputs copied into static buffers)
                gstep = GraphedStep(driver, images, tokens, masks, labels, warmup=1)
                graphed = True
            except Exception as e:                       # same HIP kernels, launched eagerly
                log(f'HIP graph capture failed ({type(e).__name__}: {e}); running the step eagerly')
                if args.graph == 'on':
                    raise

        def step():
            if gstep is not None:
                return gstep.step(images, tokens, masks, labels)
            return driver.step(images, tokens, masks, labels)

        def eager_step():
            return driver.step(images, tokens, masks, labels)
    else:
        opt = torch.optim.AdamW(groups, weight_decay=1e-4)

        def step():
            opt.zero_grad(set_to_none=True)
            out = dp.forward(images=images, texts=tokens, modality_masks=masks)
            L = dp.compute_loss(out, labels)
            L['total_loss'].backward()
            dp.reduce_grads()
            opt.step()
            return L
        eager_step = step

    log(f'model built, {args.warmup} warm-up steps')
    for _ in range(args.warmup):
        L = step()
    log(f'timing {args.steps} steps')
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        L = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # Kernel-duration pass for the roofline: the SAME K steps again, now with a HIP event pair around every launch of the
    # dominant kernels on the launch stream.  It is a second pass because timing events are not free here: each record is a
    # system-scope fence (L2 write-back), which slowed the step by 15-25 % when taken inside the throughput region.
    prof, ln_prof = [], []
    if not args.no_kernel_events:
        ops.gemm_profile_begin(); ops.ln_profile_begin()
        te = time.perf_counter()
        for _ in range(args.steps):
            eager_step()                                 # (a replayed graph launches nothing from Python: the event pass is eager)
        torch.cuda.synchronize()
        ms_with_events = (time.perf_counter() - te) / args.steps * 1e3
        prof = ops.gemm_profile_end(); ln_prof = ops.ln_profile_end()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss = float(L['total_loss'].detach())
    if world > 1:                                  # every rank evaluates the same global loss (parallel.py): check it
        lt = torch.tensor([loss, -loss], device=dev, dtype=torch.float64)
        dist.all_reduce(lt, op=dist.ReduceOp.MAX)
        loss_spread = abs(float(lt[0]) + float(lt[1]))   # max over ranks - min over ranks; reported, and loud when it is not rounding
        if loss_spread > 1e-5 * max(1.0, abs(loss)):
            log(f'WARNING: ranks disagree on the global loss by {loss_spread:.3e} (loss {loss:.6f})')
    else:
        loss_spread = 0.0
    if rank == 0:
        value = world * B * args.steps / elapsed
        res = {
            'metric': 'multimodal instances/sec (PxK, 5-modality) at 1/2/4/8 GPU; eval queries/sec',
            'value': value, 'unit': 'instances/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': model.compute_dtype, 'data': 'synthetic',
            'config': {'workload': f'P={P},K={K} per GPU, vis/nir/sk/cp 224x224 + text (T<=77), CLIP ViT-B/16 + text tower '
                                   f'random-init, MER-LoRA r={args.rank}, masks {"all-on" if args.mask_drop == 0 else args.mask_drop}, '
                                   f'{C} ids, SDM+CE, fwd+bwd+grad sanitise/clip+AdamW (reference default trainable set)',
                       'P': P, 'K': K, 'global_batch': world * B, 'lora_rank': args.rank, 'parallelism': f'dp{world}'},
            'model_tflops_per_gpu': value / world * FLOP_PER_INSTANCE / 1e12,
            'mfma_frac_whole_step': value / world * FLOP_PER_INSTANCE / 1e12 / PEAK_BF16_TFLOPS,
            'final_loss': loss, 'loss_spread_over_ranks': loss_spread, 'hip_graph': graphed,
        }
        if prof:
            fl = sum(p[0] for p in prof); ms = sum(p[2].elapsed_time(p[3]) for p in prof)
            ach = fl / (ms * 1e-3) / 1e12
            res['roofline'] = {'kernel': 'mer_gemm_kernel<128,128,2,2>', 'bound': 'mfma', 'achieved': ach, 'peak': PEAK_BF16_TFLOPS,
                               'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_TFLOPS, 'traffic': pmc_traffic('mer_gemm_kernel<128, 128, 2, 2>'),
                               'traffic_note': 'HBM bytes per launch, (2*FETCH_SIZE+WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes of this command (profiles/r01_pmc_traffic.md); null if not collected',
                               'algorithmic_bytes_per_launch_avg': sum(p[1] for p in prof) / len(prof), 'launches': len(prof),
                               'avg_launch_us': ms * 1e3 / len(prof), 'kernel_ms_per_step': ms / args.steps, 'ms_per_step_with_events': ms_with_events,
                               'flops_per_launch_avg': fl / len(prof)}
        if ln_prof:
            nb = sum(p[0] for p in ln_prof); ms = sum(p[1].elapsed_time(p[2]) for p in ln_prof)
            gbs = nb / (ms * 1e-3) / 1e9
            res['roofline_hbm'] = {'kernel': 'ln_bwd_kernel<true, false>', 'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                   'frac': gbs / PEAK_HBM_GBS, 'traffic': pmc_traffic('ln_bwd_kernel<true, false>') or pmc_traffic('ln_bwd_kernel<true>'), 'launches': len(ln_prof),
                                   'avg_launch_us': ms * 1e3 / len(ln_prof), 'algorithmic_bytes_per_launch_avg': nb / len(ln_prof)}
        if world == 1 and not args.no_retrieval:
            del opt
            torch.cuda.empty_cache()
            log(f'train: {value:.1f} instances/s; retrieval bench')
            res['retrieval'] = retrieval_bench(dev)
            res['retrieval']['encode'] = encode_bench(model, dev, model.arch)
            log(f'retrieval: {res["retrieval"]["queries_per_s"]:.0f} q/s')
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
