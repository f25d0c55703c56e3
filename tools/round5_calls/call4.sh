#!/bin/bash
# round 5, call 4: which CU / SIMD every wave ran on and for how long
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in stamps r4stamps; do
  MBAMD_LIBRARY=build_x/libhmsbeagle_$v.so timeout 120 python tools/trace_walkg.py c5
done 2>&1 | tee gpurun_out/r5c4_trace.log
