#!/bin/bash
# round 5, call 6: distinct issue priorities for the workgroups that share a CU
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in p1 s1 p1_r4; do
  echo "== $v c5"
  MBAMD_LIBRARY=build_x/libhmsbeagle_$v.so timeout 120 python tools/trace_walkg.py c5 2>&1 | grep -v '^(\|^wave\|^bench'
done 2>&1 | tee gpurun_out/r5c6_prio.log
for v in p1_r4 s1_r4; do
  echo "== $v c3"
  MBAMD_LIBRARY=build_x/libhmsbeagle_$v.so timeout 120 python tools/trace_walkg.py c3 2>&1 | grep -v '^(\|^wave\|^bench'
done 2>&1 | tee -a gpurun_out/r5c6_prio.log
