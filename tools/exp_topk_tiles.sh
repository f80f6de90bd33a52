#!/bin/bash
# filter-pass tile of the batched retrieval at 8..512 queries (REID_TOPK_TILE is read once per process)
for t in -1 4 5 6; do
  echo "REID_TOPK_TILE=$t"
  REID_TOPK_TILE=$t python tools/bench_topk_mid.py 2>/dev/null | grep -v "^{" | cut -c1-60
done
