#!/bin/bash
# round 5, call 3: split kernel with the counters as LDS accesses -- A/B and stamps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do
  timeout 120 python tools/ablate_walkg.py c5
  MBAMD_LIBRARY=build_x/libhmsbeagle_r4.so timeout 120 python tools/ablate_walkg.py c5
done 2>&1 | tee gpurun_out/r5c3_ab.log
MBAMD_LIBRARY=build_x/libhmsbeagle_stamps.so timeout 120 python tools/trace_walkg.py c5 2>&1 | tee gpurun_out/r5c3_trace.log
timeout 300 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "codon or general_state or golden or eviction or deferred or walk_waves" 2>&1 | tail -3
