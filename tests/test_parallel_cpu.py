"""CPU, gloo, world_size 2: the data-parallel layer (prcv2025reid_amd/parallel.py).

Checks, with the CPU oracle standing in for the head, that evaluating the head on ALL-GATHERED features on every
rank and taking rank-local rows of the gradient reproduces the single-process loss and gradients of the global
batch (loss identical on both ranks; feature gradients = slices; encoder-side parameter gradients add up under
all-reduce SUM)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _head_loss(raw, fmask, labels, state, arch):
    from oracle import reid_oracle as O
    sem = {m: O.sdm_module(f, state) for m, f in raw.items()}
    fused = O.feature_fusion(list(sem.values()), [fmask[m] for m in sem], state, arch['fusion_num_heads'])
    f, logits, _, _ = O.bn_neck(fused, state, True)
    out = {'logits': logits, 'feature_masks': fmask, 'raw_modality_features': raw}
    return O.compute_loss(out, labels, contrastive_weight=0.1, tau=0.2)['total_loss']


def _make(seed, B):
    g = torch.Generator().manual_seed(seed)
    mods = ('vis', 'nir', 'text')
    W = {m: torch.randn(16, 512, generator=g) * 0.3 for m in mods}          # stand-in "encoder": feat = x @ W_m
    X = {m: torch.randn(B, 16, generator=g) for m in mods}
    fmask = {'vis': torch.ones(B), 'nir': (torch.rand(B, generator=g) > 0.3).float(), 'text': torch.ones(B)}
    labels = torch.arange(B) // 2
    return mods, W, X, fmask, labels


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from collections import OrderedDict
    from prcv2025reid_amd.config import TrainingConfig, arch_of
    from prcv2025reid_amd.parallel import AllGatherRows, DataParallel, gather_no_grad
    from prcv2025reid_amd.weights import seeded_state
    arch = arch_of(TrainingConfig())
    state = {k: v for k, v in seeded_state(arch, 6, 4).items() if not k.startswith('clip_encoder.')}
    B = 8
    mods, W, X, fmask, labels = _make(0, B)
    b = B // world
    sl = slice(rank * b, (rank + 1) * b)
    Wp = {m: W[m].clone().requires_grad_(True) for m in mods}
    raw_local = OrderedDict((m, X[m][sl] @ Wp[m]) for m in mods)

    class FakeModel:                                   # just enough surface for DataParallel
        def named_parameters(self):
            return [(f'clip_encoder.fake.{m}', Wp[m]) for m in mods]
    dp = DataParallel(FakeModel())
    assert dp.world == world and dp.rank == rank
    graw, gmask = dp._gather_fn(raw_local, OrderedDict((m, fmask[m][sl]) for m in mods))
    glabels = gather_no_grad(labels[sl])
    assert torch.equal(glabels, labels) and all(torch.equal(gmask[m], fmask[m]) for m in mods)
    loss = _head_loss(graw, gmask, glabels, state, arch)
    loss.backward()
    dp.reduce_grads()
    torch.save({'loss': loss.detach(), 'grads': {m: Wp[m].grad for m in mods}}, os.path.join(tmp, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_global_head_on_gathered_features_equals_single_process(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from prcv2025reid_amd.config import TrainingConfig, arch_of
    from prcv2025reid_amd.weights import seeded_state
    arch = arch_of(TrainingConfig())
    state = {k: v for k, v in seeded_state(arch, 6, 4).items() if not k.startswith('clip_encoder.')}
    mods, W, X, fmask, labels = _make(0, 8)
    Wp = {m: W[m].clone().requires_grad_(True) for m in mods}
    raw = {m: X[m] @ Wp[m] for m in mods}
    ref = _head_loss(raw, fmask, labels, state, arch)
    ref.backward()
    outs = [torch.load(os.path.join(str(tmp_path), f'r{r}.pt')) for r in range(world)]
    for o in outs:
        assert abs(float(o['loss']) - float(ref)) < 1e-6                      # same global loss on every rank
        for m in mods:
            assert float((o['grads'][m] - Wp[m].grad).abs().max()) < 1e-5 * max(1.0, float(Wp[m].grad.abs().max()))


# ---------------------------------------------------------------------------------------------------------------------
def test_merge_topk_equals_global_ranking():
    """Per-shard top-k lists (made here with the oracle's ranking rule) merge into the global top-k, ties included."""
    from oracle import reid_oracle as O
    from prcv2025reid_amd.parallel import merge_topk
    g = torch.Generator().manual_seed(3)
    Nq, Ng, D, k, W = 17, 403, 32, 10, 4
    Q = O.l2n(torch.randn(Nq, D, generator=g)); G = O.l2n(torch.randn(Ng, D, generator=g))
    G[100] = G[7]; G[301] = G[7]; G[302] = G[7]                       # exact ties across shards
    want_idx, want_sc = O.topk_ranklist(Q, G, k)
    bounds = [0, 101, 202, 303, Ng]
    parts_i, parts_s = [], []
    for w in range(W):
        a, b = bounds[w], bounds[w + 1]
        li, ls = O.topk_ranklist(Q, G[a:b], k)
        parts_i.append(li + a); parts_s.append(ls)
    gi, gs = merge_topk(torch.stack(parts_i), torch.stack(parts_s), k)
    assert torch.equal(gi, want_idx) and torch.allclose(gs, want_sc, atol=0, rtol=0)
    # shards shorter than k are padded with -1 / -inf
    li, ls = O.topk_ranklist(Q, G[:4], 4)
    pad_i = torch.cat([li, torch.full((Nq, k - 4), -1)], 1); pad_s = torch.cat([ls, torch.full((Nq, k - 4), float('-inf'))], 1)
    li2, ls2 = O.topk_ranklist(Q, G[4:], k)
    gi, gs = merge_topk(torch.stack([pad_i, li2 + 4]), torch.stack([pad_s, ls2]), k)
    assert torch.equal(gi, want_idx)


def _shard_worker(rank, world, port):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import reid_oracle as O
    from prcv2025reid_amd import parallel as PAR
    g = torch.Generator().manual_seed(9)
    Nq, Ng, D, k = 11, 120, 16, 5
    Q = O.l2n(torch.randn(Nq, D, generator=g)); G = O.l2n(torch.randn(Ng, D, generator=g))
    a, b = rank * Ng // world, (rank + 1) * Ng // world

    class LocalIndex:                                       # the HIP GalleryIndex needs a GPU: the oracle ranks this rank's shard
        def __init__(self, Gl):
            self.Gf = Gl
        def topk(self, q, k=10, normalized=False, query_img_ids=None):
            i, s = O.topk_ranklist(q, self.Gf, k)
            return i.to(torch.int32), s
    idx = PAR.ShardedGalleryIndex.__new__(PAR.ShardedGalleryIndex)
    idx.local, idx.offset, idx.group, idx.world = LocalIndex(G[a:b]), a, None, world
    gi, gs = idx.topk(Q, k=k, normalized=True)
    want_i, want_s = O.topk_ranklist(Q, G, k)
    assert torch.equal(gi, want_i) and torch.equal(gs, want_s), rank
    dist.destroy_process_group()


def test_sharded_gallery_gloo():
    port = 29641
    mp.spawn(_shard_worker, args=(2, port), nprocs=2, join=True)


def test_bench_spawn_relays_a_dead_rank_and_exits_nonzero():
    """``bench.py --gpus 2`` (no WORLD_SIZE) on a machine whose ranks cannot start (no GPU here): the parent must notice the first dead
    rank, stop the other, relay the ranks' stderr and exit non-zero promptly -- not sit in communicate() on rank 0."""
    import subprocess
    import sys
    import time
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('needs a machine without a GPU: the ranks must fail')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--spawn-timeout', '120'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert 'rank' in r.stderr and 'stderr (tail)' in r.stderr
    assert time.time() - t0 < 120
