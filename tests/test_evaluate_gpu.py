"""GPU: on-device AP/CMC evaluator against the reference's own numbers (tests/golden/retrieval_metrics.npz, produced by
executing eval_mm_protocol.rank_and_metrics / train._reid_map) and against the CPU oracle on larger seeded cases."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from oracle import reid_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['bf16', 'f16'])
def flavor(request):
    from prcv2025reid_amd import _lib
    _lib.set_flavor(request.param)
    yield request.param
    _lib.set_flavor('bf16')


def test_metrics_golden(flavor):
    from prcv2025reid_amd.evaluate import ProtocolEvaluator
    z = np.load(os.path.join(GOLDEN, 'retrieval_metrics.npz'))
    Q = torch.tensor(z['Q']).cuda(); G = torch.tensor(z['G']).cuda()
    qp = torch.tensor(z['q_pid']); gp = torch.tensor(z['g_pid'])
    g_img = [f'g{i}' for i in range(G.shape[0])]
    q_img = [set(s.split('|')) - {''} for s in z['q_img'].tolist()]
    ev = ProtocolEvaluator(G, gp, g_img)
    r = ev.rank_and_metrics(Q, qp, q_img, ignore_same_img=True)
    assert r['num_queries'] == int(z['rm_n'])
    assert abs(r['mAP'] - float(z['rm_mAP'])) < 1e-6
    for k, kk in (('R@1', 'rm_r1'), ('R@5', 'rm_r5'), ('R@10', 'rm_r10')):
        assert abs(r[k] - float(z[kk])) < 1e-12, k
    r2 = ev.rank_and_metrics(Q, qp, q_img, ignore_same_img=False)
    assert r2['num_queries'] == int(z['rn_n']) and abs(r2['mAP'] - float(z['rn_mAP'])) < 1e-6
    for k, kk in (('R@1', 'rn_r1'), ('R@5', 'rn_r5'), ('R@10', 'rn_r10')):
        assert abs(r2[k] - float(z[kk])) < 1e-12, k
    mAP, top1 = ev.reid_map(Q, qp)
    assert abs(mAP - float(z['reid_map'])) < 1e-6 and abs(top1 - float(z['reid_top1'])) < 1e-12


def test_scores_are_fp32_grade(flavor):
    from prcv2025reid_amd.evaluate import ProtocolEvaluator
    g = torch.Generator().manual_seed(5)
    Q = torch.nn.functional.normalize(torch.randn(300, 512, generator=g), dim=1)
    G = torch.nn.functional.normalize(torch.randn(5003, 512, generator=g), dim=1)      # odd size: padded to 5004 inside
    ev = ProtocolEvaluator(G.cuda(), torch.zeros(5003, dtype=torch.long), normalized=True)
    S = ev.scores(Q.cuda(), normalized=True)[:, :5003].cpu().double()
    ref = Q.double() @ G.double().T
    assert float((S - ref).abs().max()) < 4e-7


@pytest.mark.parametrize('Nq,Ng,npid', [(64, 20000, 50), (40, 3001, 7), (16, 4100, 2)])
def test_metrics_vs_oracle(flavor, Nq, Ng, npid):
    """Clustered features (positives rank early but not first), same-image masking, pids missing from the gallery."""
    from prcv2025reid_amd.evaluate import ProtocolEvaluator
    g = torch.Generator().manual_seed(Nq + Ng)
    centers = torch.randn(npid + 3, 512, generator=g)
    gp = torch.randint(0, npid, (Ng,), generator=g)
    qp = torch.randint(0, npid + 3, (Nq,), generator=g)          # some query pids have no gallery row
    G = centers[gp] * 0.6 + torch.randn(Ng, 512, generator=g)
    Q = centers[qp] * 0.6 + torch.randn(Nq, 512, generator=g)
    g_img = [f'im{i}' for i in range(Ng)]
    q_img = []
    for i in range(Nq):
        own = [g_img[j] for j in torch.randint(0, Ng, (i % 4,), generator=g).tolist()]
        pos = (gp == qp[i]).nonzero().flatten().tolist()
        if pos and i % 2:
            own = own[:3] + [g_img[pos[0]]]
        q_img.append(own)
    want = O.rank_and_metrics(O.l2n(Q), qp, O.l2n(G), gp, q_img, g_img)
    ev = ProtocolEvaluator(G.cuda(), gp, g_img)
    got = ev.rank_and_metrics(Q.cuda(), qp, q_img)
    assert got['num_queries'] == want['num_queries']
    assert abs(got['mAP'] - want['mAP']) < 2e-6, (got, want)
    for k in ('R@1', 'R@5', 'R@10'):
        assert abs(got[k] - want[k]) < 1e-12, (k, got, want)


def test_competition_metrics_and_csv(tmp_path, flavor):
    from prcv2025reid_amd.evaluate import ProtocolEvaluator, competition_metrics
    m = competition_metrics({'single/nir': {'mAP': 0.2}, 'single/sk': {'mAP': 0.4}, 'single/cp': {'map': 0.6},
                             'single/text': 0.8, 'quad/nir+sk+cp+text': {'mAP': 0.9}})
    assert abs(m['map_single'] - 0.5) < 1e-12 and abs(m['map_avg2'] - 0.7) < 1e-12
    g = torch.Generator().manual_seed(1)
    G = torch.randn(500, 512, generator=g); Q = torch.randn(7, 512, generator=g)
    ev = ProtocolEvaluator(G.cuda(), torch.arange(500) % 9, [f'g{i}' for i in range(500)])
    out = tmp_path / 'sub.csv'
    ev.export_submission_csv(Q.cuda(), [f'q{i}' for i in range(7)], [f'g{i}' for i in range(500)], str(out), top_k=100)
    rows = out.read_text().strip().split('\n')
    assert rows[0] == 'query_key,ranked_gallery_ids' and len(rows) == 8
    want, _ = O.topk_ranklist(O.l2n(Q), O.l2n(G), 100)
    assert rows[1].split(',')[1].split(' ') == [f'g{i}' for i in want[0].tolist()]


# ------------------------------------------------------------------------------------------------------------------
# BASELINE config 4 at FULL size: 10k queries x 200k gallery x 512, top-10, through the same GalleryIndex / tile templates the
# bench uses; 1 280 of the queries (incl. every planted case) are checked against a chunked fp32 CPU ranking.
def test_config4_full_size_top10_vs_cpu_fp32():
    """Rank-list contract: the first k entries of the fp32 ranking under (score desc, index asc).  Lists are required to be
    IDENTICAL to the CPU fp32 `matmul` + stable argsort except where two scores are closer than fp32 rounding of a 512-term
    dot product (2e-7, checked in fp64) -- the reference's own argsort is unstable, so such pairs have no defined order there."""
    from prcv2025reid_amd.retrieval import GalleryIndex
    dev = torch.device('cuda', 0)
    Nq, Ng, D, k = 10000, 200000, 512, 10
    g = torch.Generator(device=dev).manual_seed(2)
    Q = torch.nn.functional.normalize(torch.randn(Nq, D, device=dev, generator=g), dim=1)
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device=dev, generator=g), dim=1)
    # planted exact ties (duplicates of a gallery row spread over the gallery; the query IS that row): index order must decide
    dup = [123, 45678, 45679, 150001, 199999]
    for j in dup[1:]:
        G[j] = G[dup[0]]
    Q[5] = G[dup[0]]
    # near-duplicates: top of the list for query 6 are 12 perturbed copies
    base = G[777].clone()
    near = torch.arange(90000, 90012, device=dev)
    G[near] = torch.nn.functional.normalize(base.view(1, -1) + 1e-3 * torch.randn(12, D, device=dev, generator=g), dim=1)
    Q[6] = base
    # same-image exclusion: query i carries image id i for i < 512; its own best match in the gallery carries the same id
    g_img = torch.full((Ng,), -1, dtype=torch.int32, device=dev)
    q_img = torch.full((Nq,), -1, dtype=torch.int32, device=dev)
    sel = torch.arange(1000, 1512, device=dev)
    G[200 + sel] = torch.nn.functional.normalize(Q[sel] + 0.05 * torch.randn(512, D, device=dev, generator=g), dim=1)   # strong positives ...
    g_img[200 + sel] = sel.to(torch.int32); q_img[sel] = sel.to(torch.int32)                                            # ... that must be masked
    index = GalleryIndex(G, normalized=True, img_ids=g_img)
    idx, sc = index.topk(Q, k=k, normalized=True, query_img_ids=q_img)
    assert int((idx < 0).sum()) == 0
    rows = torch.cat([torch.arange(0, 512), torch.arange(1000, 1512), torch.arange(9744, 10000)])          # 1 280 queries
    Qc, Gc = Q[rows.to(dev)].cpu(), G.cpu()
    excl_q = q_img[rows.to(dev)].cpu(); excl_g = g_img.cpu()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    got = idx[rows.to(dev)].cpu().long()
    bad = 0
    for a in range(0, len(rows), 256):
        sim = Qc[a:a + 256] @ Gc.t()                                            # chunked fp32 CPU GEMM
        m = (excl_q[a:a + 256].view(-1, 1) >= 0) & (excl_q[a:a + 256].view(-1, 1) == excl_g.view(1, -1))
        sim = sim.masked_fill(m, -1e9)
        ref = torch.argsort(sim, dim=1, descending=True, stable=True)[:, :k]
        same = ref == got[a:a + 256]
        if not bool(same.all()):
            for qi, r in (~same).nonzero().tolist():
                x, y = int(ref[qi, r]), int(got[a + qi, r])
                d = abs(float(Qc[a + qi].double() @ Gc[x].double()) - float(Qc[a + qi].double() @ Gc[y].double()))
                assert d < 2e-7, (a + qi, r, x, y, d)                            # an fp32-rounding near-tie, nothing else
                bad += 1
    li = rows.tolist()
    assert got[li.index(5)][:5].tolist() == sorted(dup)                          # exact ties in ascending index order
    assert set(got[li.index(6)].tolist()) <= set(near.tolist()) | {777}
    for i in range(1000, 1512, 37):                                              # the masked strong positive is absent
        assert (200 + i) not in got[li.index(i)].tolist()
    print(f'  10k x 200k top-10: {len(rows)} queries checked against CPU fp32, {bad} fp32-rounding near-tie swaps')
