#!/bin/bash
# round 5, call 8: a start stagger for the second workgroup of a CU (2048 cycles x 1, 2, 4, 8)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in g1 g2 g4 g8; do
  echo "== $v c5"
  MBAMD_LIBRARY=build_x/libhmsbeagle_$v.so timeout 120 python tools/trace_walkg.py c5 2>&1 | grep -v '^(\|^wave\|^bench\|^work'
done 2>&1 | tee gpurun_out/r5c8.log
