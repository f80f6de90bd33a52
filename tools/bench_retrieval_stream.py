"""Retrieval in the reference's streaming form (tools/eval_mm_protocol.py:401-455 ranks ONE query at a time): a few queries
against the whole 200k gallery.  Each call has to stream the 16-bit gallery copy once (Ng*D*2 bytes): HBM-bound, unlike the
batched 10k-query case (MFMA-bound).  Prints achieved gallery GB/s per query-batch size."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd.retrieval import GalleryIndex


def main():
    dev = torch.device('cuda', 0)
    Ng, D, k = 200000, 512, 10
    g = torch.Generator(device=dev).manual_seed(2)
    G = torch.nn.functional.normalize(torch.randn(Ng, D, device=dev, generator=g), dim=1)
    index = GalleryIndex(G, normalized=True)
    out = {'Ng': Ng, 'D': D, 'k': k, 'gallery_bytes_16bit': Ng * D * 2, 'gallery_bytes_fp32': Ng * D * 4, 'cases': {}}
    for Nq in (1, 2, 4, 8, 64, 128, 512):
        Q = torch.nn.functional.normalize(torch.randn(Nq, D, device=dev, generator=g), dim=1)
        for _ in range(3):
            idx, sc = index.topk(Q, k=k, normalized=True)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            idx, sc = index.topk(Q, k=k, normalized=True)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        # order of the exact (float64) scores rounded to fp32; where it differs, the two entries must be an fp32-rounding near-tie
        sim = Q.double() @ G.double().t()
        ref = torch.argsort(sim.float(), dim=1, descending=True, stable=True)[:, :k]
        same = ref == idx.long()
        worst = 0.0
        for qi, r in (~same).nonzero().tolist():
            worst = max(worst, abs(float(sim[qi, int(ref[qi, r])] - sim[qi, int(idx[qi, r])])))
        out['cases'][Nq] = {'ms': t * 1e3, 'queries_per_s': Nq / t, 'form': 'stream (fp32 gallery, 1 pass per 4 queries)' if Nq <= 4 else 'batched (16-bit gallery)',
                            'gallery_GBps': (Ng * D * 4 * ((Nq + 3) // 4) if Nq <= 4 else Ng * D * 2) / t / 1e9,
                            'identical_to_f64_order': bool(same.all()), 'largest_score_gap_where_different': worst}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
