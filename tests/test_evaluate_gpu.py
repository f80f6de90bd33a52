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
