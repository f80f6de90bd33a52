for g in 4 8 16 32 64; do echo "GROUP_M=$g"; REID_GEMM_GROUPM=$g TILES=0 python tools/bench_gemm_variants.py 2>/dev/null | sed 's/tile0: //'; done
