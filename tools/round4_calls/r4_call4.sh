#!/bin/bash
# Round 4, GPU call 4: k_walkg_s with two workgroups per CU (one LDS slot per wave beyond 32 states, older values through HBM)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call4.log; : > $OUT
say() { echo "$@" | tee -a $OUT; }
abl() { local label=$1 cfg=$2; shift 2; env "$@" MBAMD_VERBOSE=1 timeout 300 python tools/ablate_walkg.py $cfg 2>/tmp/abl.err | tail -1 | sed "s/product/$label/" | tee -a $OUT; grep -m1 "walk plan" /tmp/abl.err | tee -a $OUT; }
timeout 1500 python -m pytest tests/test_engine_gpu.py -q -m gpu --maxfail=10 > gpurun_out/r4_pytest_gpu4.log 2>&1; say "engine gpu tests exit $?"; tail -6 gpurun_out/r4_pytest_gpu4.log | tee -a $OUT
say "== c5"
abl default c5 X=1
for w in 2 3 5 6 8; do abl bins$w c5 MBAMD_WALK_WAVES=$w; done
abl bins2_slots4 c5 MBAMD_WALK_WAVES=2 MBAMD_MAX_LDS_SLOTS=4
abl bins4_slots2 c5 MBAMD_WALK_WAVES=4 MBAMD_MAX_LDS_SLOTS=2
abl G2 c5 MBAMD_WALKG_G=2
abl G2_bins4 c5 MBAMD_WALKG_G=2 MBAMD_WALK_WAVES=4
abl old_kernel c5 MBAMD_WALKG_SHARED=0
say "== c3"
abl default c3 X=1
for w in 3 4; do abl bins$w c3 MBAMD_WALK_WAVES=$w; done
abl G4 c3 MBAMD_WALKG_G=4
abl G4_bins3 c3 MBAMD_WALKG_G=4 MBAMD_WALK_WAVES=3
abl slots2 c3 MBAMD_MAX_LDS_SLOTS=2
abl old_kernel c3 MBAMD_WALKG_SHARED=0
for m in m3 wag; do
  say "== trace $m"
  MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_wgs_trace.so timeout 300 python tools/trace_walkgs.py $m 2>&1 | tail -14 | tee -a $OUT
done
