#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 with no WORLD_SIZE in the environment: this process stays CPU-only, SPAWNS N rank processes
(one per GPU, torch.distributed backend nccl = RCCL over xGMI, rendezvous on 127.0.0.1) and relays rank 0's JSON line; when the
driver launches the ranks itself (python -m torch.distributed.run ... bench.py --gpus N) WORLD_SIZE is set and must equal N.

A "step" = one pass of the training hot path over one synthetic P x K batch per GPU:
zero_grad -> forward (4 vision modalities + text through the HIP executor) -> CE + SDM losses ->
backward (hand-written HIP backward, LoRA/bn_neck/null-token gradients = the reference's default
trainable set, train.py:1418-1425) -> gradient all-reduce (N > 1) -> AdamW step.
Workload: N = 1: BASELINE.json configs[1] (P=16, K=4, LoRA r=8, masks all-on, ViT-B/16 + CLIP text random-init, 400
identities).  N > 1: configs[2] (P=32, K=4 PER GPU, same model; weak scaling), and the line also carries the single-GPU
value of that same per-GPU workload measured on rank 0 alone (`n1_same_workload`).  Inputs are resident in HBM.

One JSON line on stdout (rank 0).  Besides the contract keys it carries
  roofline      -- dominant kernel (the MER GEMM: mer_gemm_pps_kernel / mer_gemm_pp_kernel / mer_gemm_kernel<128,128,2,2>, bf16 MFMA): algorithmic FLOPs of every launch in
                   the timed region / its duration measured with HIP events on the launch stream
  flavors       -- step time of BOTH 16-bit operand flavors (bf16 = headline, f16) on the same workload
  parity        -- per flavor, HIP vs the CPU oracle on the full-size batch of this workload (regularisers off): per-modality
                   features, bn_features / 8, the three losses -- so the record shows which flavor meets 1e-3 and its speed
  cpu_baseline  -- the CPU oracle (oracle/reid_oracle.py, kind "port") timed on this host on a bounded sample, plus the two
                   CPU retrieval baselines of SURVEY 8(d)
  retrieval     -- eval queries/s of the fused cosine top-10 on 10k x 200k x 512 (BASELINE.json configs[3])
"""
import argparse
import json
import os
import socket
import subprocess
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
    ap.add_argument('--P', type=int, default=None, help='identities per GPU (default: 16 at N=1 = BASELINE config 2, 32 at N>1 = config 3)')
    ap.add_argument('--K', type=int, default=4)
    ap.add_argument('--rank', type=int, default=8, help='LoRA rank')
    ap.add_argument('--mask-drop', type=float, default=0.0)
    ap.add_argument('--accum', type=int, default=1,
                    help='gradient-accumulation steps: a bench "step" stays one micro-batch (forward + losses + backward); the all-reduce and the '
                         'optimizer run every ACCUM-th step (reference: accum = 16 // min(8, P*K) = 2, train.py:1364,1482-1483; BASELINE configs[4])')
    ap.add_argument('--spawn-timeout', type=float, default=1500.0, help='--gpus N without WORLD_SIZE: overall deadline (s) for the spawned ranks')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-retrieval', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--optimizer', default='fused', choices=['fused', 'torch'])
    ap.add_argument('--graph', default='off', choices=['auto', 'on', 'off'],
                    help='replay the step as a HIP graph (single process only; measured ~1 ms/step SLOWER than eager launches on MI355X, r01)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL); gloo only for rehearsals')
    ap.add_argument('--compute-dtype', default=None, choices=[None, 'bf16', 'f16'], help='headline flavor (default bf16)')
    ap.add_argument('--no-second-flavor', action='store_true', help='skip the step-time leg of the other 16-bit flavor')
    ap.add_argument('--no-parity', action='store_true', help='skip the in-run full-size parity check against the CPU oracle')
    return ap.parse_args()


PMC_FILE = 'r04_pmc_traffic.json'
# kernels the roofline objects are about: the committed PMC summary must have been collected on a tree that dispatches kernels of
# these names, or the traffic figure is refused (null + a note) instead of silently describing other code
GEMM_KERNELS = ('mer_gemm_pps_kernel', 'mer_gemm_pp_kernel', 'mer_gemm_kernel<128, 128, 2, 2')
LN_KERNELS = ('ln_bwd8_kernel',)


def pmc_traffic(prefixes, required=None):
    """(launch-weighted HBM bytes per launch of the kernels whose name starts with one of `prefixes`, note) from the committed PMC summary
    (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, corrected as the microarch guide prescribes;
    tools/pmc_collect.sh).  Every name of `required` must occur in the file: a summary made before a kernel was renamed or replaced no
    longer describes the dispatched code and is refused."""
    path = os.path.join(ROOT, 'profiles', PMC_FILE)
    if not os.path.exists(path):
        return None, f'profiles/{PMC_FILE} not collected'
    d = json.load(open(path))
    missing = [r for r in (required or prefixes) if not any(k.startswith(r) for k in d)]
    if missing:
        return None, f'profiles/{PMC_FILE} has no kernel named {missing}: stale summary refused'
    ks = [k for k in d if any(k.startswith(p) for p in prefixes)]
    n = sum(d[k]['launches'] for k in ks)
    if n == 0:
        return None, 'no launches'
    return sum(d[k]['traffic_bytes_per_launch'] * d[k]['launches'] for k in ks) / n, \
        f'HBM bytes per launch, (2*FETCH_SIZE+WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes of this command (profiles/{PMC_FILE})'


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
    res = {'value': 8.0 / t, 'unit': 'instances/s', 'cores': torch.get_num_threads(), 'kind': 'port',
           'sample': 'oracle fp32, P=4,K=2 (8 instances), LoRA r=4, fwd+loss+bwd, 1 warm-up + 3 timed steps',
           'seconds_per_step': t}
    res['retrieval'] = cpu_retrieval_baselines(cores)
    return res


def cpu_retrieval_baselines(cores):
    """The two CPU retrieval baselines of SURVEY 8(d) on the 200k x 512 gallery (oracle functions, host cores):
    (a) chunked fp32 Q @ G.T + topk(10) over 1 000 queries; (b) the reference's form -- one query at a time, GEMV + FULL
    argsort of the 200k scores (tools/eval_mm_protocol.py:401-423, train.py:450-479) -- over 100 queries."""
    from oracle import reid_oracle as O
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(2)
    Ng, D = 200000, 512
    G = torch.nn.functional.normalize(torch.randn(Ng, D, generator=g), dim=1)
    Q = torch.nn.functional.normalize(torch.randn(1000, D, generator=g), dim=1)
    log('cpu_baseline: chunked fp32 GEMM + top-10, 1000 queries x 200k')
    O.cosine_sim(Q[:100], G).topk(10, dim=1)
    t0 = time.perf_counter()
    O.cosine_sim(Q, G).topk(10, dim=1)
    ta = time.perf_counter() - t0
    log('cpu_baseline: reference-style per-query GEMV + full argsort, 100 queries x 200k')
    O.rank_full(O.cosine_sim(Q[:1], G).squeeze(0))
    t0 = time.perf_counter()
    for i in range(100):
        O.rank_full(O.cosine_sim(Q[i:i + 1], G).squeeze(0))
    tb = time.perf_counter() - t0
    return {'chunked_gemm_topk10': {'queries_per_s': 1000 / ta, 'sample': '1000 queries x 200k x 512 fp32 matmul + topk(10)', 'cores': cores},
            'per_query_gemv_full_argsort': {'queries_per_s': 100 / tb, 'sample': '100 queries x 200k x 512, one GEMV + full stable argsort each '
                                            '(eval_mm_protocol.py:401-423)', 'cores': cores}}


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


def visible_gpu_count():
    """GPUs this process would see, counted WITHOUT initialising HIP (the spawning parent must stay GPU-free: a process that has
    touched the GPU must not start others on this pool): the visibility lists if set, else the KFD topology in sysfs (nodes with
    SIMDs are GPUs).  None when neither source is readable -- the ranks then validate their own device index."""
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(',') if x.strip() != ''])
    base = '/sys/class/kfd/kfd/topology/nodes'
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, 'properties')) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            if int(props.get('simd_count', '0')) > 0:
                n += 1
        return n
    except OSError:
        return None


def spawn_ranks(args):
    """N > 1 and no WORLD_SIZE: this (parent) process never touches a GPU; it starts N fresh rank processes, WATCHES ALL OF THEM
    and relays rank 0's line.  The first rank that exits non-zero, or the overall deadline, ends the run: the other ranks are
    killed (a rank stuck in a collective whose peer died would otherwise hang until the caller's own timeout), every rank's stderr
    tail is relayed and the parent exits non-zero."""
    import tempfile
    n = args.gpus
    ndev = visible_gpu_count()
    if args.backend == 'nccl' and ndev is not None and ndev < n:
        raise SystemExit(f'bench.py --gpus {n}: only {ndev} GPU(s) visible (use --backend gloo to rehearse several ranks on one GPU)')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs, logs = [], []
    out0 = tempfile.TemporaryFile(mode='w+')
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        err = tempfile.TemporaryFile(mode='w+')
        logs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=err, text=True))
    deadline = time.monotonic() + args.spawn_timeout
    failed = None
    while True:
        rcs = [pr.poll() for pr in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = f'rank {bad[0]} exited with code {rcs[bad[0]]}'
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            failed = f'deadline of {args.spawn_timeout:.0f} s passed (exit codes so far {rcs})'
            break
        time.sleep(0.2)
    if failed:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        for pr in procs:
            pr.wait()
    out0.seek(0)
    out = out0.read()
    line = [ln for ln in out.splitlines() if ln.startswith('{')]
    if failed or not line:
        for r, err in enumerate(logs):
            err.seek(0)
            tail = err.read()[-4000:]
            sys.stderr.write(f'---- rank {r} stderr (tail) ----\n{tail}\n')
        sys.stderr.write(out)
        raise SystemExit(f'bench.py: {failed or "rank 0 printed no result line"}; exit codes {[pr.returncode for pr in procs]}')
    for r, err in enumerate(logs):                         # relay rank 0's progress lines
        if r == 0:
            err.seek(0)
            sys.stderr.write(err.read())
    print(line[-1], flush=True)


def build_model(local, rank_lora, flavor, C, regularisers=True):
    from prcv2025reid_amd.config import TrainingConfig
    from prcv2025reid_amd.model import CLIPBasedMultiModalReIDModel, apply_reference_freeze
    kw = {} if regularisers else dict(drop_path=0.0, modality_dropout=0.0, dropout_rate=0.0, fusion_dropout=0.0, sdm_dropout=0.0)
    # init='seeded': every tensor random (lora_B included), so no path of the step is vacuous
    cfg = TrainingConfig(device=f'cuda:{local}', mer_lora_rank=rank_lora, contrastive_weight=0.1, seed=0, compute_dtype=flavor,
                         init='seeded', **kw)
    model = CLIPBasedMultiModalReIDModel(cfg)
    model.set_num_classes(C)
    apply_reference_freeze(model)
    model.set_epoch(2)
    model.train()
    return model


def make_stepper(model, dp, args, images, tokens, masks, labels, world):
    groups = [dict(params=[p for p in g['params'] if p.requires_grad], lr=g['lr'], name=g['name']) for g in model.get_learnable_params()]
    groups = [g for g in groups if g['params']]
    graphed = False
    if args.optimizer == 'fused':
        # the reference's step (train.py:975-1047): sanitise, adaptive clip, AdamW -- three fused launches, no host sync
        from prcv2025reid_amd.trainer import FusedAdamW, StepDriver, GraphedStep
        opt = FusedAdamW(groups, weight_decay=1e-4)
        driver = StepDriver(dp, opt, accum_steps=max(1, args.accum), adaptive_clip=True, dp=dp)
        gstep = None
        if world == 1 and args.graph != 'off' and args.accum == 1:
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

        if args.accum > 1:
            raise SystemExit('--accum needs the fused optimizer path (StepDriver)')

        def step():
            opt.zero_grad(set_to_none=True)
            out = dp.forward(images=images, texts=tokens, modality_masks=masks)
            L = dp.compute_loss(out, labels)
            L['total_loss'].backward()
            dp.reduce_grads()
            opt.step()
            return L
        eager_step = step
    return step, eager_step, graphed


def timed(step, steps, warmup, world, dev):
    import torch.distributed as dist
    for _ in range(warmup):
        L = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        L = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    return elapsed, L


def head_section_ms(model, B, C, dev, reps=5):
    """Time of the head section alone (model.head_section + compute_loss + their backward) on a batch of B samples: under data parallelism
    every rank evaluates it on the GLOBAL batch (parallel.py), so this is the redundant work per rank that does not shrink with the world
    size -- reported so that a scaling result below the target can be attributed."""
    from collections import OrderedDict
    g = torch.Generator(device=dev).manual_seed(5)
    mods = list(model.arch['modalities'])
    raw = OrderedDict((m, torch.randn(B, model.arch['fusion_dim'], device=dev, generator=g).requires_grad_(True)) for m in mods)
    fmask = OrderedDict((m, torch.ones(B, device=dev)) for m in mods)
    labels = (torch.arange(B, device=dev) // 4) % C
    was = model.training
    model.train()

    def once():
        out = model.head_section(raw, fmask)
        L = model.compute_loss(out, labels)
        gr = torch.autograd.grad(L['total_loss'], list(raw.values()), allow_unused=True)
        return gr

    for _ in range(2):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    model.train(was)
    return (time.perf_counter() - t0) / reps * 1e3


def parity_check(local, rank_lora, C, P, K, flavors):
    """HIP (each flavor) vs the CPU oracle on the FULL-SIZE batch of the benchmarked workload, training forward with the random
    regularisers off (the oracle is deterministic): unit-normalised per-modality features and bn_features / 8, the three losses."""
    from oracle import reid_oracle as O
    from prcv2025reid_amd.synthetic import synthetic_batch
    from prcv2025reid_amd.weights import seeded_state
    res = {'oracle': 'oracle/reid_oracle.py (fp32 CPU restatement pinned to the reference fixtures), train-mode forward + losses, '
                     f'regularisers off, same seeded weights and batch (P={P},K={K}, r={rank_lora}, masks all-on)', 'tolerance_north_star': 1e-3}
    ref = Lr = None
    for fl in flavors:
        model = build_model(local, rank_lora, fl, C, regularisers=False)
        arch = model.arch
        if ref is None:
            state = seeded_state(arch, C, 0)
            batch = synthetic_batch(P, K, arch, seed=1000, num_classes=C)
            tok = model.tokenizer(batch['texts'], return_tensors='pt', padding=True, truncation=True, max_length=77)
            cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1), 16))
            torch.set_num_threads(cores)
            log(f'parity: oracle forward on {P * K} instances ({cores} threads)')
            t0 = time.perf_counter()
            with torch.no_grad():
                ref = O.forward(state, arch, batch['images'], tok, batch['modality_mask'], True)
                Lr = O.compute_loss(ref, batch['person_id'], contrastive_weight=0.1, tau=0.2)
            res['oracle_seconds'] = time.perf_counter() - t0
        model.load_state_dict(state)
        with torch.no_grad():
            out = model(images={m: t.to(f'cuda:{local}') for m, t in batch['images'].items()},
                        texts={k: v.to(f'cuda:{local}') for k, v in tok.items()}, modality_masks=batch['modality_mask'])
            L = model.compute_loss(out, batch['person_id'].to(f'cuda:{local}'))
        emb = float((out['bn_features'].cpu() / 8 - ref['bn_features'] / 8).abs().max())
        per = {}
        for m in ref['raw_modality_features']:
            a = torch.nn.functional.normalize(out['raw_modality_features'][m].cpu(), dim=1)
            b = torch.nn.functional.normalize(ref['raw_modality_features'][m], dim=1)
            per[m] = float((a - b).abs().max())
        dl = {k: abs(float(L[k]) - float(Lr[k])) for k in ('total_loss', 'ce_loss', 'sdm_loss')}
        worst = max([emb] + list(per.values()) + list(dl.values()))
        res[fl] = {'bn_features_unit_maxabs': emb, 'per_modality_unit_maxabs': per, 'loss_abs': dl,
                   'losses_hip': {k: float(L[k]) for k in dl}, 'losses_oracle': {k: float(Lr[k]) for k in dl},
                   'worst': worst, 'meets_north_star_1e-3': bool(worst <= 1e-3)}
        log(f'parity[{fl}]: bn_features/8 {emb:.2e}, per-modality {max(per.values()):.2e}, losses {max(dl.values()):.2e}')
        del model
        torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        return spawn_ranks(args)
    world = int(env_world or '1')
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU '
                         f'(python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}) or drop WORLD_SIZE')
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
    from prcv2025reid_amd.parallel import DataParallel
    from prcv2025reid_amd.synthetic import synthetic_batch

    C = 400
    P = args.P if args.P is not None else (16 if world == 1 else 32)        # BASELINE configs[1] / configs[2]
    K = args.K
    B = P * K
    head = args.compute_dtype or 'bf16'
    other = 'f16' if head == 'bf16' else 'bf16'
    model = build_model(local, args.rank, head, C)
    dp = DataParallel(model)
    batch = synthetic_batch(P, K, model.arch, seed=1000 + rank, mask_drop=args.mask_drop, num_classes=C, label_offset=rank * P)
    images = {m: t.to(dev) for m, t in batch['images'].items()}
    masks = batch['modality_mask']                      # host tensors, as the reference's collate produces them
    tok = model.tokenizer(batch['texts'], return_tensors='pt', padding=True, truncation=True, max_length=77)
    tokens = {k: v.to(dev) for k, v in tok.items()}     # pre-tokenised, resident in HBM
    labels = batch['person_id'].to(dev)
    step, eager_step, graphed = make_stepper(model, dp, args, images, tokens, masks, labels, world)

    log(f'[{head}] model built, {args.warmup} warm-up + {args.steps} timed steps (P={P},K={K} per GPU, world {world})')
    elapsed, L = timed(step, args.steps, args.warmup, world, dev)
    # Kernel-duration pass for the roofline: the SAME K steps again, now with a HIP event pair around every launch of the
    # dominant kernels on the launch stream.  It is a second pass because timing events are not free here: each record is a
    # system-scope fence (L2 write-back), which slowed the step by 15-25 % when taken inside the throughput region.
    prof, ln_prof = [], []
    ms_with_events = None
    if not args.no_kernel_events:
        ops.gemm_profile_begin(); ops.ln_profile_begin()
        te = time.perf_counter()
        for _ in range(args.steps):
            eager_step()                                 # (a replayed graph launches nothing from Python: the event pass is eager)
        torch.cuda.synchronize()
        ms_with_events = (time.perf_counter() - te) / args.steps * 1e3
        prof = ops.gemm_profile_end(); ln_prof = ops.ln_profile_end()
    loss = float(L['total_loss'].detach())
    loss_spread = params_spread = 0.0
    if world > 1:                                  # every rank evaluates the same global loss (parallel.py): check it
        lt = torch.tensor([loss, -loss], device=dev, dtype=torch.float64)
        dist.all_reduce(lt, op=dist.ReduceOp.MAX)
        loss_spread = abs(float(lt[0]) + float(lt[1]))   # max over ranks - min over ranks; reported, and loud when it is not rounding
        if loss_spread > 1e-5 * max(1.0, abs(loss)):
            log(f'WARNING: ranks disagree on the global loss by {loss_spread:.3e} (loss {loss:.6f})')
        params_spread = dp.params_in_sync()              # replicas must stay bit-identical (head gradients are averaged too)
    value = world * B * args.steps / elapsed
    res = {
        'metric': 'multimodal instances/sec (PxK, 5-modality) at 1/2/4/8 GPU; eval queries/sec',
        'value': value, 'unit': 'instances/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': model.compute_dtype, 'data': 'synthetic',
        'config': {'workload': f'P={P},K={K} per GPU, vis/nir/sk/cp 224x224 + text (T<=77), CLIP ViT-B/16 + text tower '
                               f'random-init, MER-LoRA r={args.rank}, masks {"all-on" if args.mask_drop == 0 else args.mask_drop}, '
                               f'{C} ids, SDM+CE, fwd+bwd+grad sanitise/clip+AdamW (reference default trainable set)'
                               + (f', gradient accumulation {args.accum} (optimizer every {args.accum} steps)' if args.accum > 1 else ''),
                   'baseline_config': 'configs[1] 1xMI355X P=16,K=4' if (world == 1 and P == 16) else
                                      ('configs[2] DP P=32,K=4 per GPU' if (P == 32 and args.rank == 8 and args.mask_drop == 0) else
                                       ('configs[4] stress P=64,K=4 r=16 mask-drop 30% accum 2 (per-GPU slice)'
                                        if (P == 64 and args.rank == 16 and args.accum == 2 and abs(args.mask_drop - 0.3) < 1e-9) else 'custom')),
                   'P': P, 'K': K, 'global_batch': world * B, 'lora_rank': args.rank, 'accum_steps': args.accum, 'mask_drop': args.mask_drop,
                   'parallelism': f'dp{world}'},
        'model_tflops_per_gpu': value / world * FLOP_PER_INSTANCE / 1e12,
        'mfma_frac_whole_step': value / world * FLOP_PER_INSTANCE / 1e12 / PEAK_BF16_TFLOPS,
        'final_loss': loss, 'loss_spread_over_ranks': loss_spread, 'params_spread_over_ranks': params_spread,
        'rccl_world': (world if (world > 1 and args.backend == 'nccl') else 0), 'backend': args.backend if world > 1 else None,
        'hip_graph': graphed,
    }
    if prof:
        fl = sum(p[0] for p in prof); ms = sum(p[2].elapsed_time(p[3]) for p in prof)
        ach = fl / (ms * 1e-3) / 1e12
        traffic, tnote = pmc_traffic(GEMM_KERNELS)
        # what an event pair measures around NOTHING on this stream (marker-to-marker time): the part of every launch's figure that is
        # not the kernel.  Reported beside the raw numbers, never subtracted from `achieved` / `frac`.
        pairs = []
        torch.cuda.synchronize()
        for _ in range(64):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record(); pairs.append((e0, e1))
        torch.cuda.synchronize()
        empty_us = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)[len(pairs) // 2]
        net_ms = max(ms - empty_us * 1e-3 * len(prof), 1e-9)
        res['roofline'] = {'kernel': 'the MER GEMM: mer_gemm_pps_kernel<EPI, BM> (persistent 256x256 / 224x256 ping-pong tiles: plain and residual epilogues), '
                                     'mer_gemm_pp_kernel<EPI, BM> (same tiles, one per workgroup: GELU / multiply-by-derivative epilogues), '
                                     'mer_gemm_kernel<128,128,2,2,EPI> (short-K narrow-N shapes); merged per-modality weights, row groups',
                           'bound': 'mfma', 'achieved': ach, 'peak': PEAK_BF16_TFLOPS,
                           'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_TFLOPS, 'traffic': traffic, 'traffic_note': tnote,
                           'algorithmic_bytes_per_launch_avg': sum(p[1] for p in prof) / len(prof), 'launches': len(prof),
                           'avg_launch_us': ms * 1e3 / len(prof), 'kernel_ms_per_step': ms / args.steps, 'ms_per_step_with_events': ms_with_events,
                           'flops_per_launch_avg': fl / len(prof),
                           'empty_event_pair_us': empty_us,
                           'frac_net_of_event_pairs': fl / (net_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                           'event_note': 'every launch is bracketed by its own event pair inside the running step; an EMPTY pair on the same stream '
                                         'measures empty_event_pair_us, and rocprofv3 of the same command (profiles/) gives the kernels\' own durations: '
                                         '`frac` is the raw event figure, `frac_net_of_event_pairs` the same with the empty-pair time taken off every launch',
                           'flops_counted': '2*M*N*K per launch (merged weights: no low-rank K extension, no padded columns)',
                           'clock_note': 'in-kernel shader clock under this load 1.86-1.90 GHz (profiles/r03_gemm_clock.log): the dense bf16 MFMA rate '
                                         'at that clock is 1.96 PFLOP/s; `peak` stays the 2.4 GHz figure of the microarch guide'}
    if ln_prof:
        nb = sum(p[0] for p in ln_prof); ms = sum(p[1].elapsed_time(p[2]) for p in ln_prof)
        gbs = nb / (ms * 1e-3) / 1e9
        traffic, tnote = pmc_traffic(LN_KERNELS)
        res['roofline_hbm'] = {'kernel': 'ln_bwd8_kernel (16-bit cotangent in, residual-stream gradient in IEEE half, eight columns per lane)', 'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                               'frac': gbs / PEAK_HBM_GBS, 'traffic': traffic, 'traffic_note': tnote, 'launches': len(ln_prof),
                               'avg_launch_us': ms * 1e3 / len(ln_prof), 'algorithmic_bytes_per_launch_avg': nb / len(ln_prof),
                               'note': 'event pairs around the launches inside the running step (the side stream shares the chip); since r04 the kernel '
                                       'moves 12 B per element (residual-stream gradient in IEEE half) instead of 16: 81-84 us per launch alone against 106 '
                                       'for the r03 form, i.e. a shorter kernel at a lower fraction of the HBM peak'}
    res['flavors'] = {head: {'ms_per_step': elapsed / args.steps * 1e3, 'value': value, 'role': 'headline'}}
    if rank == 0 and not args.no_kernel_events:             # (profiled runs pass --no-kernel-events: their kernel trace is of the step only)
        # the head section alone at this rank's batch and at the global batches of configs 3 / 5: under DP every rank runs it on the GLOBAL
        # batch, so (head at world * B) - (head at B) is per-rank work that data parallelism adds
        try:
            hs = {str(b): head_section_ms(model, b, C, dev) for b in sorted({B, world * B, 8 * 128})}
            res['head_section_ms'] = {'by_batch': hs, 'local_batch': B, 'global_batch': world * B,
                                      'what': 'head_section + compute_loss + backward to the encoder features, wall time per call, this GPU alone'}
        except Exception as e:                              # (attribution only: never fail the bench line on it)
            res['head_section_ms'] = {'error': repr(e)}
    if world > 1:
        # single-GPU value of the SAME per-GPU workload, rank 0 alone (the others wait): what the N-GPU value is to be divided by
        dist.barrier()
        if rank == 0:
            s1, _, _ = make_stepper(model, DataParallel(model, enabled=False), args, images, tokens, masks, labels, 1)
            e1, _ = timed(s1, args.steps, 1, 1, dev)
            res['n1_same_workload'] = {'value': B * args.steps / e1, 'ms_per_step': e1 / args.steps * 1e3, 'unit': 'instances/s',
                                       'what': f'P={P},K={K} on one GPU (rank 0 alone, no collectives), same process and model'}
            res['scaling_vs_n1_same_workload'] = value / res['n1_same_workload']['value']
        dist.barrier()
    del step, eager_step
    if world == 1 and not args.no_second_flavor:
        # the other 16-bit flavor on the same workload, same K / W (f16 meets north_star's 1e-3 as written; see `parity`)
        del dp, model
        torch.cuda.empty_cache()
        m2 = build_model(local, args.rank, other, C)
        dp2 = DataParallel(m2)
        s2, _, _ = make_stepper(m2, dp2, args, images, tokens, masks, labels, 1)
        log(f'[{other}] second flavor: {args.warmup} warm-up + {args.steps} timed steps')
        e2, L2 = timed(s2, args.steps, args.warmup, 1, dev)
        res['flavors'][other] = {'ms_per_step': e2 / args.steps * 1e3, 'value': B * args.steps / e2, 'final_loss': float(L2['total_loss'].detach())}
        del s2, dp2
        model = m2
    if rank == 0:
        if world == 1 and not args.no_parity:
            torch.cuda.empty_cache()
            res['parity'] = parity_check(local, args.rank, C, P, K, [head, other])
            for fl in res['flavors']:                    # which of the two step times belongs to a result inside north_star's tolerance
                if fl in res['parity']:
                    res['flavors'][fl]['meets_north_star_1e-3'] = res['parity'][fl]['meets_north_star_1e-3']
        if world == 1 and not args.no_retrieval:
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
