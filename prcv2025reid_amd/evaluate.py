"""Batched MM-protocol evaluator on the HIP path.

Reference: ``rank_and_metrics`` (tools/eval_mm_protocol.py:369-469), ``_reid_map`` (train.py:450-479), the aggregation of
``validate_competition_style`` (train.py:591-602) and ``export_submission_csv`` (tools/eval_mm_protocol.py:595-649).

The reference scores one query at a time, argsorts the whole gallery row and walks it in Python.  Here
  * scores come from the MFMA GEMM of the hot path with SPLIT operands: every fp32 feature is written as a sum of 16-bit
    pieces (2 for the f16 flavor, 3 for bf16) and the significant cross products are laid side by side along K, so one
    ``reid_mer_gemm`` call returns fp32-grade similarities (error <= ~2e-7 for unit vectors, i.e. fp32 rounding level)
    at matrix-core speed;
  * AP / CMC come from ``reid_rank_metrics`` (csrc/metrics.hip): ranks of the positives only, no sort of the gallery.
Nothing here falls back to the CPU; the pure-Python oracle lives in oracle/reid_oracle.py and is used by tests only.
"""
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib, ops
from .retrieval import GalleryIndex, l2_normalize

_SCALE = 16.0        # features are scaled into the 16-bit formats' comfortable range before splitting (undone by alpha)


def _split(x: torch.Tensor, pieces: int) -> List[torch.Tensor]:
    out, r = [], x
    for _ in range(pieces):
        h = ops.to_t16(r)
        out.append(h)
        r = r - h.float()
    return out


def _split_operands(Qf: torch.Tensor, Gf: torch.Tensor):
    """[Q pieces laid along K], [G pieces laid along K] such that Qcat @ Gcat.T ~= (Q @ G.T) * SCALE^2 to fp32 accuracy."""
    f16 = _lib.flavor() == 'f16'
    n = 2 if f16 else 3
    pairs = [(0, 0), (0, 1), (1, 0)] if f16 else [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
    q = _split(Qf * _SCALE, n); g = _split(Gf * _SCALE, n)
    return torch.cat([q[i] for i, _ in pairs], 1).contiguous(), torch.cat([g[j] for _, j in pairs], 1).contiguous()


class ProtocolEvaluator:
    """Gallery-side state built once: normalised features, split 16-bit operand, pid CSR, image ids."""

    def __init__(self, gallery_feats: torch.Tensor, gallery_pids: torch.Tensor, gallery_img_ids: Optional[Sequence] = None,
                 normalized: bool = False):
        g = gallery_feats.contiguous().float()
        if not g.is_cuda:
            raise _lib.ReidHipError('ProtocolEvaluator needs device tensors (there is no CPU path)')
        self.dev = g.device
        self.Gf = g if normalized else l2_normalize(g)
        self.Ng, self.D = self.Gf.shape
        pad = (-self.Ng) % 4
        Gp = torch.cat([self.Gf, torch.zeros(pad, self.D, device=self.dev)], 0) if pad else self.Gf
        self._Gcat = _split_operands(Gp[:1], Gp)[1]          # only the gallery half is kept
        pids = gallery_pids.to(self.dev).long()
        self.g_pid = pids.to(torch.int32).contiguous()
        uniq, inv = torch.unique(pids, return_inverse=True)
        order = torch.argsort(inv, stable=True)               # gallery rows grouped by pid, ascending row inside a group
        counts = torch.bincount(inv, minlength=uniq.numel())
        self.csr_off = torch.cat([torch.zeros(1, dtype=torch.long, device=self.dev), counts.cumsum(0)]).to(torch.int32).contiguous()
        self.csr_idx = order.to(torch.int32).contiguous()
        self.max_pos = int(counts.max())
        self._uniq = uniq
        self._img_map: Dict = {}
        self.g_img = None
        if gallery_img_ids is not None:
            ids = [self._img_id(x) for x in gallery_img_ids]
            self.g_img = torch.tensor(ids, dtype=torch.int32, device=self.dev)
        self.index = None

    def _img_id(self, x) -> int:
        if x is None:
            return -1
        return self._img_map.setdefault(x, len(self._img_map))

    # ---------------------------------------------------------------------------------------------------------
    def scores(self, q_feats: torch.Tensor, normalized: bool = False) -> torch.Tensor:
        """fp32-grade cosine similarities [nq, ld >= Ng] (ld a multiple of 4; columns >= Ng are padding)."""
        Qf = q_feats.contiguous().float().to(self.dev)
        if not normalized:
            Qf = l2_normalize(Qf)
        Qcat = _split_operands(Qf, Qf[:1])[0]
        S = torch.empty(Qf.shape[0], self._Gcat.shape[0], device=self.dev)
        ops.gemm(Qcat, self._Gcat, S, alpha=1.0 / (_SCALE * _SCALE))
        return S

    def per_query(self, q_feats: torch.Tensor, q_pids: torch.Tensor, q_img_ids: Optional[Sequence] = None,
                  ignore_same_img: bool = True, chunk: int = 1024, normalized: bool = False):
        """(ap f64 [Nq], rank1 i32 [Nq], npos i32 [Nq]) on the device."""
        Nq = q_feats.shape[0]
        qp = q_pids.to(self.dev).long()
        pos = torch.searchsorted(self._uniq, qp).clamp(max=self._uniq.numel() - 1)
        slot = torch.where(self._uniq[pos] == qp, pos, torch.full_like(pos, -1)).to(torch.int32).contiguous()
        qp32 = qp.to(torch.int32).contiguous()
        excl = None
        if ignore_same_img and q_img_ids is not None and self.g_img is not None:
            rows = []
            for ids in q_img_ids:
                ids = ids if isinstance(ids, (set, list, tuple)) else [ids]
                known = [self._img_map[x] for x in ids if x is not None and x in self._img_map]   # unknown ids mask nothing
                if len(known) > 4:
                    raise ValueError('at most 4 image ids per query (one per modality sample)')
                rows.append(known + [-1] * (4 - len(known)))
            excl = torch.tensor(rows, dtype=torch.int32, device=self.dev)
        ap = torch.zeros(Nq, dtype=torch.float64, device=self.dev)
        rank1 = torch.zeros(Nq, dtype=torch.int32, device=self.dev)
        npos = torch.zeros(Nq, dtype=torch.int32, device=self.dev)
        for a in range(0, Nq, chunk):
            b = min(Nq, a + chunk)
            S = self.scores(q_feats[a:b], normalized)
            ops.rank_metrics(S, self.g_pid, self.g_img, qp32[a:b], slot[a:b], None if excl is None else excl[a:b].contiguous(),
                             self.csr_off, self.csr_idx, self.Ng, self.max_pos, ap[a:b], rank1[a:b], npos[a:b])
        return ap, rank1, npos

    def rank_and_metrics(self, q_feats, q_pids, q_img_ids=None, ignore_same_img: bool = True, chunk: int = 1024) -> Dict[str, float]:
        """Same dictionary as eval_mm_protocol.py:455-469: queries without an (unmasked) positive are skipped."""
        ap, rank1, npos = self.per_query(q_feats, q_pids, q_img_ids, ignore_same_img, chunk)
        if bool((npos < 0).any()):
            raise _lib.ReidHipError('a query has more than 8192 positives in the gallery: not supported by reid_rank_metrics')
        valid = npos > 0
        n = int(valid.sum())
        if n == 0:
            return {'mAP': 0.0, 'R@1': 0.0, 'R@5': 0.0, 'R@10': 0.0, 'num_queries': 0}
        r = rank1[valid]
        return {'mAP': float(ap[valid].mean()), 'R@1': float((r <= 1).double().mean()), 'R@5': float((r <= 5).double().mean()),
                'R@10': float((r <= 10).double().mean()), 'num_queries': n}

    def reid_map(self, q_feats, q_pids):
        """(mAP, top-1) of _reid_map (train.py:450-479): mAP over queries with a positive, top-1 over ALL queries."""
        ap, rank1, npos = self.per_query(q_feats, q_pids, None, False)
        valid = npos > 0
        n = max(1, int(valid.sum()))
        return float(ap[valid].sum() / n), float(((rank1 == 1) & valid).double().sum() / q_feats.shape[0])

    # ---------------------------------------------------------------------------------------------------------
    def export_submission_csv(self, q_feats, query_keys: Sequence[str], gallery_img_names: Sequence, output_path: str,
                              top_k: int = 100):
        """eval_mm_protocol.py:595-649: one row per query, the top_k gallery image ids of the unmasked ranking."""
        import csv
        if self.index is None:
            self.index = GalleryIndex(self.Gf, normalized=True)
        idx, _ = self.index.topk(q_feats.to(self.dev), k=min(top_k, self.Ng))
        idx = idx.cpu().tolist()
        with open(output_path, 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['query_key', 'ranked_gallery_ids'])
            for key, row in zip(query_keys, idx):
                w.writerow([key, ' '.join(str(gallery_img_names[i]) for i in row if gallery_img_names[i] is not None)])


def competition_metrics(all_metrics: Dict[str, Dict[str, float]]) -> Dict[str, float]:
    """train.py:578-602: mean of the four single-modality mAPs, the four-modality mAP and their average."""
    def _get_map(m):
        if isinstance(m, dict):
            for k in ('mAP', 'map', 'mAP_mean', 'map_mean'):
                if k in m:
                    return float(m[k])
        if isinstance(m, (int, float)):
            return float(m)
        return 0.0
    singles = [_get_map(all_metrics.get(k, {})) for k in ('single/nir', 'single/sk', 'single/cp', 'single/text')]
    map_single = sum(singles) / max(1, len([x for x in singles if x == x]))
    map_quad = _get_map(all_metrics.get('quad/nir+sk+cp+text', {}))
    return {'map_single': map_single, 'map_quad': map_quad, 'map_avg2': (map_single + map_quad) / 2.0}
