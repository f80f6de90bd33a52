"""Pure data parallelism over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-process; this layer is new.  The P x K batch is split over ranks
(whole identities per rank); the encoders run on local rows only.  Three batch-level reductions couple the
samples (BN-neck batch statistics models/model.py:216, the CE mean :546, the SDM similarity matrices
:586-622), so the five [B, 512] per-modality features, their masks and the labels are ALL-GATHERED
(~1.3 MB per rank at B=128) and the tiny head + losses are evaluated on the GLOBAL batch by every rank,
redundantly and identically.  Consequences:
  * the loss equals the single-process reference's loss on the global batch (same BN statistics, same
    negatives for SDM), not an average of per-rank losses;
  * no collective is needed in backward for the gather: rank r's feature gradient is rows r of the
    (replicated) global gradient;
  * head parameters (bn_neck, sdm_module, fusion) get the same gradient on every rank up to the summation order of the
    kernels' fp32 atomics; they are averaged in the small bucket so the replicas stay bit-identical (reduce_grads);
  * encoder-side parameters (the flat LoRA arena, null tokens) hold per-rank partial sums -> ONE
    all-reduce(SUM) of the flat arena (21 MB fp32 at r=8) + one tiny one for the null tokens.
    xGMI rings are per-link bound: t ~ 2(n-1)/n * 21 MB / 153 GB/s ~ 0.24 ms at n=8, <1 % of a step, so a
    single bucket after backward is used (nothing to overlap with).
"""
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


class AllGatherRows(torch.autograd.Function):
    """out = concat over ranks of x (dim 0).  Backward = this rank's rows of the output gradient; valid because
    every rank evaluates the same function of the gathered tensor (see module docstring)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, group):
        world = dist.get_world_size(group)
        ctx.rank = dist.get_rank(group)
        ctx.rows = x.shape[0]
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows].contiguous(), None


def gather_no_grad(x: torch.Tensor, group=None) -> torch.Tensor:
    world = dist.get_world_size(group)
    x = x.contiguous()
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out


class DataParallel:
    """Wraps a CLIPBasedMultiModalReIDModel for one-process-per-GPU data parallel training.

    usage per step:
        out = dp.forward(images, texts, masks)            # head runs on the global batch
        loss = dp.compute_loss(out, labels)               # labels of this rank; gathered inside
        loss['total_loss'].backward(); dp.reduce_grads(); optimizer.step()
    Every rank must pass the same set of modalities (keys of ``images`` / presence of ``texts``).
    """

    def __init__(self, model, group=None, enabled: bool = True):
        """``enabled=False``: behave as a single process even when torch.distributed is initialised (a rank running alone)."""
        self.model = model
        self.group = group
        on = enabled and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0

    def _gather_fn(self, raw: "OrderedDict[str, torch.Tensor]", fmask: "OrderedDict[str, torch.Tensor]"):
        if self.world == 1:
            return raw, fmask
        # ONE collective for all modality features ([B, n_mod*D]) and one for all masks ([B, n_mod]): a small RCCL call costs
        # tens of microseconds of latency whatever its size, and there would be 2*n_mod of them per step
        names = list(raw.keys())
        D = raw[names[0]].shape[1]
        g = AllGatherRows.apply(torch.cat([raw[m] for m in names], dim=1), self.group)
        graw = OrderedDict((m, g[:, i * D:(i + 1) * D]) for i, m in enumerate(names))
        mnames = list(fmask.keys())
        gm = gather_no_grad(torch.stack([fmask[m].float() for m in mnames], dim=1), self.group)
        gmask = OrderedDict((m, gm[:, i].contiguous()) for i, m in enumerate(mnames))
        return graw, gmask

    def forward(self, images=None, texts=None, modality_masks=None, return_features=False):
        return self.model(images=images, texts=texts, modality_masks=modality_masks, return_features=return_features,
                          gather_fn=self._gather_fn)

    def compute_loss(self, outputs, labels):
        if self.world > 1:
            labels = gather_no_grad(labels.to(outputs['logits'].device), self.group)
        return self.model.compute_loss(outputs, labels)

    def local_rows(self, t: torch.Tensor) -> torch.Tensor:
        b = t.shape[0] // self.world
        return t[self.rank * b:(self.rank + 1) * b]

    def encoder_side_params(self):
        """Parameters whose gradients are per-rank partial sums."""
        return [p for n, p in self.model.named_parameters()
                if p.requires_grad and (n.startswith('clip_encoder.') or n.startswith('null_tokens.'))]

    def head_params(self):
        """Parameters of the redundantly evaluated head (bn_neck, sdm_module, feature_fusion)."""
        return [p for n, p in self.model.named_parameters()
                if p.requires_grad and not (n.startswith('clip_encoder.') or n.startswith('null_tokens.'))]

    def _small_bucket(self):
        """ONE flat fp32 buffer holding the gradients of every small trainable tensor (null tokens first, then the head), with each
        ``p.grad`` a VIEW into it: the step then needs two collectives (LoRA arena, this bucket) and no cat / copy-back of ~40
        tensors.  Rebuilt (current values copied in) whenever somebody replaced a ``.grad`` tensor."""
        small = [p for p in self.encoder_side_params() if p.numel() < (1 << 16)]
        n_sum = len(small)
        small += self.head_params()
        b = getattr(self, '_bucket', None)
        ok = b is not None and len(b['params']) == len(small) and all(a is c for a, c in zip(b['params'], small)) and \
            all(p.grad is not None and p.grad.data_ptr() == ptr for p, ptr in zip(small, b['ptrs']))
        if ok:
            return b
        offs, o = [], 0
        for p in small:
            offs.append(o)
            o += (p.numel() + 3) // 4 * 4                      # 16-byte aligned slices (FusedAdamW's requirement)
        dev = small[0].device if small else torch.device('cpu')
        flat = torch.zeros(max(o, 4), dtype=torch.float32, device=dev)
        ptrs = []
        for p, off in zip(small, offs):
            view = flat[off:off + p.numel()].view_as(p)
            if p.grad is not None:
                view.copy_(p.grad)
            p.grad = view
            ptrs.append(view.data_ptr())
        head_off = offs[n_sum] if n_sum < len(small) else o
        self._bucket = dict(params=small, ptrs=ptrs, flat=flat, head=flat[head_off:o], n=o)
        return self._bucket

    def reduce_grads(self):
        """Encoder-side gradients are per-rank partial sums -> all-reduce(SUM).  Head gradients are mathematically identical
        on every rank, but the head kernels reduce with fp32 atomics (BN-neck statistics, LayerNorm dgamma/dbeta), so their
        last bits differ from rank to rank; left alone the replicas would drift apart (Adam amplifies differences on
        near-zero gradients).  They are therefore AVERAGED: every rank steps with the same bits.
        Two collectives per step: the flat LoRA arena (its own bucket) and the flat small-tensor bucket (null tokens summed,
        head pre-scaled by 1 / world so that the SUM is the average)."""
        if self.world == 1:
            return
        b = self._small_bucket()
        for p in self.encoder_side_params():
            if p.numel() >= (1 << 16):
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)     # the flat LoRA arena: one bucket
        if b['n'] > 0:
            if b['head'].numel():
                b['head'].mul_(1.0 / self.world)
            dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group)

    def params_in_sync(self) -> float:
        """max over parameters of (max over ranks - min over ranks) of the parameter values: 0.0 when the replicas agree
        bit for bit (diagnostic for tests and the bench line)."""
        if self.world == 1:
            return 0.0
        flat = torch.cat([p.detach().reshape(-1).double() for _, p in self.model.named_parameters() if p.requires_grad])
        hi = flat.clone(); lo = flat.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        return float((hi - lo).abs().max())


# --------------------------------------------------------------------------------------------------------------------
# Sharded gallery retrieval: gallery rows split over ranks, queries replicated
def merge_topk(idx_parts: torch.Tensor, score_parts: torch.Tensor, k: int):
    """Merge W per-shard top-k lists into the global top-k under the ranking rule of the single-GPU path
    (score descending, GLOBAL gallery index ascending on ties).

    idx_parts int32/int64 [W, Nq, k] (global indices, -1 = no entry), score_parts f32 [W, Nq, k] -> (idx [Nq, k], score [Nq, k]).
    Exact: every member of the global top-k is in the top-k of the shard that holds it.
    Pure index bookkeeping on k*W candidates per query (the heavy part, scoring the shard, is the HIP kernel)."""
    W, Nq, kk = idx_parts.shape
    idx = idx_parts.permute(1, 0, 2).reshape(Nq, W * kk).long()
    sc = score_parts.permute(1, 0, 2).reshape(Nq, W * kk).float()
    sc = torch.where(idx < 0, torch.full_like(sc, float('-inf')), sc)
    # two stable sorts = lexicographic (score desc, index asc)
    o1 = torch.argsort(torch.where(idx < 0, torch.full_like(idx, 1 << 62), idx), dim=1, stable=True)
    idx1, sc1 = idx.gather(1, o1), sc.gather(1, o1)
    o2 = torch.argsort(sc1, dim=1, descending=True, stable=True)[:, :k]
    return idx1.gather(1, o2), sc1.gather(1, o2)


class ShardedGalleryIndex:
    """Rank r holds gallery rows [r*Ng/W, (r+1)*Ng/W); ``topk`` runs the local fused cosine top-k (retrieval.GalleryIndex),
    all-gathers the k*W candidates (k*8 bytes per query per rank) and merges them -- the global fp32 ranking, exactly."""

    def __init__(self, local_gallery: torch.Tensor, shard_offset: int, group=None, normalized: bool = False, img_ids=None):
        from .retrieval import GalleryIndex
        self.local = GalleryIndex(local_gallery, normalized=normalized, img_ids=img_ids)
        self.offset = int(shard_offset)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def topk(self, queries: torch.Tensor, k: int = 10, normalized: bool = False, query_img_ids=None):
        kk = min(k, self.local.Gf.shape[0])
        idx, sc = self.local.topk(queries, k=kk, normalized=normalized, query_img_ids=query_img_ids)
        idx = torch.where(idx >= 0, idx.long() + self.offset, idx.long())
        if kk < k:                                            # a shard smaller than k: pad with empty entries
            pad = k - kk
            idx = torch.cat([idx, torch.full((idx.shape[0], pad), -1, dtype=idx.dtype, device=idx.device)], 1)
            sc = torch.cat([sc, torch.full((sc.shape[0], pad), float('-inf'), device=sc.device)], 1)
        if self.world == 1:
            return merge_topk(idx.unsqueeze(0), sc.unsqueeze(0), k)
        gi = gather_no_grad(idx.unsqueeze(0).contiguous(), self.group)
        gs = gather_no_grad(sc.unsqueeze(0).contiguous(), self.group)
        return merge_topk(gi, gs, k)
