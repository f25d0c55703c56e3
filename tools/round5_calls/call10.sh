#!/bin/bash
# round 5, call 10: k_walkg2 with distinct idle loads
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "codon or general_state or golden or eviction or deferred or walk_waves or state_counts" 2>&1 | tail -3
for c in c5 c3; do
  timeout 120 python tools/ablate_walkg.py $c
  MBAMD_LIBRARY=build_x/libhmsbeagle_r4.so timeout 120 python tools/ablate_walkg.py $c
done 2>&1 | tee gpurun_out/r5c10_ab.log
for c in c5 c3; do
  echo "== w2 $c"
  MBAMD_LIBRARY=build_x/libhmsbeagle_w2.so timeout 120 python tools/trace_walkg.py $c 2>&1 | grep -v '^(\|^wave\|^bench'
done 2>&1 | tee gpurun_out/r5c10_trace.log
