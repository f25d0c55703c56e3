#!/bin/bash
# round 5, call 7: two accumulator sets per child (no dependent MFMA issue) with and without distinct priorities
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in p2 p2_noprio s1; do
  echo "== $v c5"
  MBAMD_LIBRARY=build_x/libhmsbeagle_$v.so timeout 120 python tools/trace_walkg.py c5 2>&1 | grep -v '^(\|^wave\|^bench'
done 2>&1 | tee gpurun_out/r5c7.log
