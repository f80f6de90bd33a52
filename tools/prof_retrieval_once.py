"""One retrieval call under rocprofv3 (kernel stats), optionally with REID_TOPK_TILE / REID_TOPK_DBG."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prcv2025reid_amd.retrieval import GalleryIndex
g = torch.Generator(device='cuda').manual_seed(2)
Q = torch.nn.functional.normalize(torch.randn(10000, 512, device='cuda', generator=g), dim=1)
G = torch.nn.functional.normalize(torch.randn(200000, 512, device='cuda', generator=g), dim=1)
ix = GalleryIndex(G, normalized=True)
for _ in range(3):
    ix.topk(Q, k=10, normalized=True)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ix.topk(Q, k=10, normalized=True)
e1.record(); torch.cuda.synchronize()
print(f'tile={os.environ.get("REID_TOPK_TILE")} dbg={os.environ.get("REID_TOPK_DBG")}: {e0.elapsed_time(e1) / 5:.3f} ms per call')
