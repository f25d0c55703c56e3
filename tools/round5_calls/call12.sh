#!/bin/bash
# round 5, call 12: k_walkg2 -- swapped bins in the second workgroup of a CU, table touch two entries ahead
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in w2 b_swap b_touch b_both; do
  echo "== $v c5"
  MBAMD_LIBRARY=build_x/libhmsbeagle_$v.so timeout 120 python tools/trace_walkg.py c5 2>&1 | grep '^kernel\|^CUs with'
done 2>&1 | tee gpurun_out/r5c12.log
for v in w2 b_touch; do
  echo "== $v c3"
  MBAMD_LIBRARY=build_x/libhmsbeagle_$v.so timeout 120 python tools/trace_walkg.py c3 2>&1 | grep '^kernel\|^CUs with'
done 2>&1 | tee -a gpurun_out/r5c12.log
