#!/bin/bash
# call 34: does the fullest SIMD set k_walk4_t's time at DNA 1000 x 50 000 (3 128 waves on 1 024 SIMDs)?  Pattern counts around it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c34; export TMPDIR=/tmp
timeout 1500 python tools/wave_tail.py 60 2>&1 | tee gpurun_out/c34/wave_tail.txt
