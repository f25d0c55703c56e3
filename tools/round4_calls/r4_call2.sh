#!/bin/bash
# Round 4, GPU call 2: where does k_walkg_s spend its time?  in-kernel clock stamps + ablation builds (wrong values on purpose)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call2.log; : > $OUT
say() { echo "$@" | tee -a $OUT; }
for m in m3 wag; do
  say "== trace $m"
  MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_wgs_trace.so timeout 300 python tools/trace_walkgs.py $m 2>&1 | tail -14 | tee -a $OUT
done
say "== ablations (kernel ms per evaluation)"
for cfg in c5 c3; do
  for v in product wgs_plain wgs_nostore wgs_nowait wgs_nostore_nowait wgs_nobar wgs_nodma wgs_notip; do
    if [ $v = product ]; then timeout 300 python tools/ablate_walkg.py $cfg 2>&1 | tail -1 | tee -a $OUT
    else MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_$v.so timeout 300 python tools/ablate_walkg.py $cfg 2>&1 | tail -1 | tee -a $OUT; fi
  done
  MBAMD_WALKG_SHARED=0 timeout 300 python tools/ablate_walkg.py $cfg 2>&1 | tail -1 | sed 's/product/old kernel/' | tee -a $OUT
done
say "== geometry after the slot-rule fix"
for cfg in c5 c3; do
  for g in 2 4; do MBAMD_WALKG_G=$g timeout 300 python tools/ablate_walkg.py $cfg 2>&1 | tail -1 | sed "s/product/G=$g/" | tee -a $OUT; done
done
