#!/bin/bash
# round 5, call 11: ablations of k_walkg2 (C5, then C3 for a few)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in w2 a_nosync a_nofetch a_notiny a_nomfma a_nostore a_nokeep a_floor a_onlymfma; do
  echo "== $v c5"
  MBAMD_LIBRARY=build_x/libhmsbeagle_$v.so timeout 120 python tools/trace_walkg.py c5 2>&1 | grep '^kernel\|^CUs with'
done 2>&1 | tee gpurun_out/r5c11_ablate.log
for v in a_nofetch a_nomfma a_nostore a_floor a_onlymfma; do
  echo "== $v c3"
  MBAMD_LIBRARY=build_x/libhmsbeagle_$v.so timeout 120 python tools/trace_walkg.py c3 2>&1 | grep '^kernel\|^CUs with'
done 2>&1 | tee -a gpurun_out/r5c11_ablate.log
