#!/bin/bash
# round 5, call 2: clock stamps by entry type and phase, split kernel against round 4's
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in stamps r4stamps; do
  MBAMD_LIBRARY=build_x/libhmsbeagle_$v.so timeout 120 python tools/trace_walkg.py c5
done 2>&1 | tee gpurun_out/r5c2_trace.log
MBAMD_LIBRARY=build_x/libhmsbeagle_r4stamps.so timeout 120 python tools/trace_walkg.py c3 2>&1 | tee -a gpurun_out/r5c2_trace.log
