#!/bin/bash
# round 5, call 14: k_walkg2 with the operands requested in front of the stores, one and a half entries ahead
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "codon or general_state or golden or eviction or deferred or walk_waves or state_counts" 2>&1 | tail -3
for c in c5 c3; do
  timeout 120 python tools/ablate_walkg.py $c
  MBAMD_LIBRARY=build_x/libhmsbeagle_r4.so timeout 120 python tools/ablate_walkg.py $c
done 2>&1 | tee gpurun_out/r5c14_ab.log
echo "== st2 c5"
MBAMD_LIBRARY=build_x/libhmsbeagle_st2.so timeout 120 python tools/trace_walkg.py c5 2>&1 | grep -v '^(\|^wave\|^bench\|^w[0-9]' | tee gpurun_out/r5c14_trace.log
