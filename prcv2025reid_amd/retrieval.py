"""Cosine gallery retrieval on the HIP path (reference: train.py:428-501, tools/eval_mm_protocol.py:50-53,
369-469, 595-649).

``cosine_topk`` is the replacement for the reference's ``sim = q @ G.T; torch.argsort(sim, descending=True)``
when only the first k ranks are consumed (CMC@1/5/10, the top-100 submission lists): it returns exactly the
first k entries of the fp32 ranking under the (score desc, index asc) rule without materialising the
[Nq, Ng] similarity matrix.  ``l2_normalize`` is ``F.normalize(x.float(), dim=1)`` (train.py:442).
Everything runs in libreid_hip.so; there is no CPU path.
"""
from typing import Optional, Tuple

import torch

from . import ops


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    x = x.contiguous().float()
    y = torch.empty_like(x)
    ops.l2norm_rows(x, y=y, eps=eps)
    return y


class GalleryIndex:
    """Device-resident gallery: fp32 rows (for exact re-scoring) + 16-bit copy (MFMA operand)."""

    def __init__(self, gallery: torch.Tensor, normalized: bool = False, img_ids: Optional[torch.Tensor] = None):
        g = gallery.contiguous().float()
        self.Gf = g if normalized else l2_normalize(g)
        self.Gb = ops.to_t16(self.Gf)
        self.img_ids = None if img_ids is None else img_ids.to(self.Gf.device, torch.int32).contiguous()
        self._ws = None
        self._ws_stream = None
        self._exact_scratch = None
        self._slots = None             # device list of flagged queries of the last large call: [count, ids...]
        # Large calls resolve candidate-list overflows (queries flagged idx[q, 0] = -2) on the device, without reading the flags back:
        # the exact pass owns one scratch row of Ng floats per QUERY of the call (untouched unless the query is flagged), so a call is cut
        # into query chunks whose scratch stays below this many bytes (10k x 200k: 8 GB of the 288 GB, one chunk).
        self.exact_scratch_bytes = 16 << 30

    def topk(self, queries: torch.Tensor, k: int = 10, normalized: bool = False,
             query_img_ids: Optional[torch.Tensor] = None, stream: Optional[bool] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(indices int32 [Nq, k], scores f32 [Nq, k]); entries whose img id equals the query's are excluded
        (same-image mask of eval_mm_protocol.py:421-422) when both id vectors are given.  ``stream``: force the one-pass
        form for a few queries (True) or the batched MFMA pipeline (False); default: whichever fits the shape.
        Index -1 = no entry (fewer than k gallery rows); no other negative value is ever returned."""
        Qf = queries.contiguous().float()
        if not normalized:
            Qf = l2_normalize(Qf)
        Nq, Ng = Qf.shape[0], self.Gf.shape[0]
        exq = exg = None
        if query_img_ids is not None and self.img_ids is not None:
            exq = query_img_ids.to(Qf.device, torch.int32).contiguous(); exg = self.img_ids
        if stream is None:
            # one query: the one-pass fp32 form (78 us at 200k x 512); 2-4 queries: the batched entry point's query-resident scan where it
            # applies (83-88 us against 91-105 us for the fp32 passes), else the fp32 form
            stream = ops.topk_stream_ok(Nq, Ng, Qf.shape[1], k) and (Nq == 1 or not ops.topk_scan_ok(Nq, Ng, Qf.shape[1], k))
        if stream:
            # a handful of queries: one pass over the fp32 gallery (the reference's per-query form), no host sync
            need = ops.topk_stream_ws_bytes(k)             # (its own buffer, zero-filled once: it holds the arrival counter of the fused merge)
            if self._ws_stream is None or self._ws_stream.numel() != need:
                self._ws_stream = torch.zeros(need, dtype=torch.uint8, device=Qf.device)
            idx = torch.empty(Nq, k, dtype=torch.int32, device=Qf.device)
            sc = torch.empty(Nq, k, dtype=torch.float32, device=Qf.device)
            ops.cosine_topk_stream(Qf, self.Gf, k, self._ws_stream, idx, sc, exclude_q=exq, exclude_g=exg)
            return idx, sc
        chunk = max(1, int(self.exact_scratch_bytes // (4 * Ng)))
        if Nq > chunk and Nq * Ng > (1 << 26):
            idx = torch.empty(Nq, k, dtype=torch.int32, device=Qf.device)
            sc = torch.empty(Nq, k, dtype=torch.float32, device=Qf.device)
            for q0 in range(0, Nq, chunk):
                i_, s_ = self._topk_batched(Qf[q0:q0 + chunk], k, None if exq is None else exq[q0:q0 + chunk], exg)
                idx[q0:q0 + chunk] = i_; sc[q0:q0 + chunk] = s_
            return idx, sc
        return self._topk_batched(Qf, k, exq, exg)

    def _topk_batched(self, Qf, k, exq, exg):
        Nq, Ng = Qf.shape[0], self.Gf.shape[0]
        Qb = ops.to_t16(Qf)
        need = ops.topk_ws_bytes(Nq, Ng, k)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=Qf.device)
        idx = torch.empty(Nq, k, dtype=torch.int32, device=Qf.device)
        sc = torch.empty(Nq, k, dtype=torch.float32, device=Qf.device)
        ops.cosine_topk(Qb, self.Gb, Qf, self.Gf, k, self._ws, idx, sc, exclude_q=exq, exclude_g=exg)
        if Nq * Ng <= (1 << 26):       # small problems: run the exact pass unconditionally -- its kernels return at once for queries
            # that are not flagged -- instead of reading the flags back (a host sync costs more than two empty launches here)
            scratch = torch.empty(Nq * Ng, dtype=torch.float32, device=Qf.device)
            ops.cosine_topk_exact(Qf, self.Gf, k, scratch, idx, sc, exclude_q=exq, exclude_g=exg)
            return idx, sc
        # Large problems: the flagged queries are compacted into a device list and ALL of them take the exact fp32 pass -- three small
        # launches, nothing read back (r02 synchronised here on every call; r03 resolved at most 256 and left the rest marked).
        need = Nq * Ng
        if self._exact_scratch is None or self._exact_scratch.numel() < need:
            self._exact_scratch = None
            self._exact_scratch = torch.empty(need, dtype=torch.float32, device=Qf.device)
        if self._slots is None or self._slots.numel() < Nq + 1:
            self._slots = torch.empty(Nq + 1, dtype=torch.int32, device=Qf.device)
        ops.cosine_topk_exact_slots(Qf, self.Gf, k, Nq, self._slots, self._exact_scratch, idx, sc, exclude_q=exq, exclude_g=exg)
        return idx, sc


def cmc_from_topk(topk_idx: torch.Tensor, q_pids: torch.Tensor, g_pids: torch.Tensor, ks=(1, 5, 10)):
    """CMC@k over queries that have a positive in the gallery (eval_mm_protocol.py:425-441).  "Has a positive" is a lookup in the
    gallery's identity histogram ([max pid + 1] counts), not an [Nq, Ng] comparison (2 GB at 10k x 200k)."""
    dev = topk_idx.device
    gp = g_pids.to(dev).long(); qp = q_pids.to(dev).long()
    if bool((topk_idx < -1).any()):                       # (this function synchronises anyway: the metrics go to the host)
        raise ValueError('cmc_from_topk: rank list holds an unresolved marker (index < -1)')
    hit = gp[topk_idx.long().clamp_min(0)] == qp.view(-1, 1)
    hit &= topk_idx >= 0                                  # (-1 = no entry: fewer than k gallery rows)
    lo = int(min(int(gp.min()), int(qp.min()))) if gp.numel() and qp.numel() else 0
    counts = torch.bincount(gp - lo, minlength=int(max(int(gp.max()), int(qp.max())) - lo + 1)) if gp.numel() else torch.zeros(1, device=dev)
    has_pos = counts[qp - lo] > 0
    n_pos = int(has_pos.sum())
    out = {}
    for k in ks:
        out[f'R@{k}'] = float(hit[:, :k].any(dim=1)[has_pos].float().mean()) if n_pos > 0 else 0.0
    return out
