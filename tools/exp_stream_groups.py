"""One query x 200k x 512 (fp32 gallery, one-pass form): wall time per GalleryIndex.topk call for several workgroup counts (knob STREAM_GROUPS)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd import _lib
from prcv2025reid_amd.retrieval import GalleryIndex
dev = torch.device('cuda', 0)
Ng, D, k = 200000, 512, 10
g = torch.Generator(device=dev).manual_seed(2)
G = torch.nn.functional.normalize(torch.randn(Ng, D, device=dev, generator=g), dim=1)
index = GalleryIndex(G, normalized=True)
Q = torch.nn.functional.normalize(torch.randn(1, D, device=dev, generator=g), dim=1)
for rnd in range(2):
    for groups in (-1, 128, 192, 256, 384, 512, 768):
        _lib.check(_lib.lib().reid_set_knob(b'STREAM_GROUPS', groups))
        for _ in range(5): index.topk(Q, k=k, normalized=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): index.topk(Q, k=k, normalized=True)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 200
        print(f'round {rnd} groups {groups:4d}: {t * 1e6:6.1f} us per call = {Ng * D * 4 / t / 1e9:6.0f} GB/s = {Ng * D * 4 / t / 8e12 * 100:4.1f} % of 8 TB/s', flush=True)
