#!/bin/bash
# round 5, call 5: ablations of the split kernel -- wave durations on CUs with one / two workgroups
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in s1 s1_r4 s1_nostore s1_notiny s1_nofetch s1_nosync s1_nomfma s1_nofetch_nostore; do
  echo "== $v"
  MBAMD_LIBRARY=build_x/libhmsbeagle_$v.so timeout 120 python tools/trace_walkg.py c5 2>&1 | grep -v '^(\|^wave\|^bench'
done 2>&1 | tee gpurun_out/r5c5_ablate.log
