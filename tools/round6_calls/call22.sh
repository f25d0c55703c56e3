#!/bin/bash
# call 22: what a full-tree evaluation costs between two rescalings of a beaglescaling=dynamic chain (SCALE_READ entries: every
# operation divides by stored exponents) against beaglescaling=always (SCALE_WRITE), DNA 500 x 20 000 and 1000 x 50 000
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c22; export TMPDIR=/tmp
{ timeout 600 python tools/scale_read_time.py bench_c2 200; timeout 900 python tools/scale_read_time.py bench_c4 60; } 2>&1 | tee gpurun_out/c22/scale_read.txt
