#!/bin/bash
# Round 4, GPU call 6: epilogues that scale by multiplication (both general-state kernels); k_walkg_s with tip chunks gathered from
# staged gather tables; k_walkg with a unified register file beyond 32 states
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call6.log; : > $OUT
say() { echo "$@" | tee -a $OUT; }
abl() { local label=$1 cfg=$2; shift 2; env "$@" MBAMD_VERBOSE=1 timeout 300 python tools/ablate_walkg.py $cfg 2>/tmp/abl.err | tail -1 | sed "s/product/$label/" | tee -a $OUT; }
timeout 1500 python -m pytest tests/test_engine_gpu.py -q -m gpu --maxfail=10 > gpurun_out/r4_pytest_gpu6.log 2>&1; say "engine gpu tests exit $?"; tail -6 gpurun_out/r4_pytest_gpu6.log | tee -a $OUT
for cfg in c5 c3; do
  say "== $cfg"
  abl walkg_r4 $cfg X=1
  abl walkg_r4_unified_regs $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_wg_uni.so
  abl walkg_r3 $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_r3.so
  abl shared $cfg MBAMD_WALKG_SHARED=1
  for w in 2 3 5; do abl shared_bins$w $cfg MBAMD_WALKG_SHARED=1 MBAMD_WALK_WAVES=$w; done
  abl shared_G4_bins3 $cfg MBAMD_WALKG_SHARED=1 MBAMD_WALKG_G=4 MBAMD_WALK_WAVES=3
  abl shared_G2_bins3 $cfg MBAMD_WALKG_SHARED=1 MBAMD_WALKG_G=2 MBAMD_WALK_WAVES=3
  abl walkg_r4_again $cfg X=1
done
for m in m3 wag; do
  say "== trace $m (k_walkg_s)"
  MBAMD_WALKG_SHARED=1 MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_wgs_trace.so timeout 300 python tools/trace_walkgs.py $m 2>&1 | tail -16 | tee -a $OUT
done
