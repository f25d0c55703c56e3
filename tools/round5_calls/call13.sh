#!/bin/bash
# round 5, call 13: k_walkg2 phase stamps, waves alone on a CU against waves sharing a SIMD
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for c in c5 c3; do
  echo "== st2 $c"
  MBAMD_LIBRARY=build_x/libhmsbeagle_st2.so timeout 120 python tools/trace_walkg.py $c 2>&1 | grep -v '^(\|^wave\|^bench\|^w[0-9]'
done 2>&1 | tee gpurun_out/r5c13.log
