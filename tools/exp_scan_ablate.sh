#!/bin/bash
# ablations of scan::scan_filter_kernel (trace build: tools/build_variant.sh scantrace -DREID_SCAN_TRACE): REID_SCAN_DBG 1 = no scoring, 2 = no publish / bar requests, 4 = no enqueue
for d in ${@:-0 1 2}; do echo "== REID_SCAN_DBG=$d"; REID_SCAN_DBG=$d REID_LIB_BF16=prcv2025reid_amd/csrc/libreid_hip_scantrace.so timeout -k 10 100 python tools/exp_scan_trace.py 5 128 2>&1 | grep -E "queries|prologue|main loop|step 0 landed|revisit|polls"; done
