#!/bin/bash
# Round 4, GPU call 9: k_walkg geometry now that 61 states fit two waves per SIMD (unified register file): bins x slots
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call9.log; : > $OUT
abl() { local label=$1 cfg=$2; shift 2; env "$@" MBAMD_VERBOSE=1 timeout 300 python tools/ablate_walkg.py $cfg 2>/tmp/abl.err | tail -1 | sed "s/product/$label/" | tee -a $OUT; grep -m1 "walk plan" /tmp/abl.err | cut -c1-160 | tee -a $OUT; }
for cfg in c5 c3; do
  echo "== $cfg" | tee -a $OUT
  abl default $cfg X=1
  abl W4_slots2 $cfg MBAMD_WALK_WAVES=4 MBAMD_MAX_LDS_SLOTS=2
  abl W4_slots3 $cfg MBAMD_WALK_WAVES=4 MBAMD_MAX_LDS_SLOTS=3
  abl W4_slots4 $cfg MBAMD_WALK_WAVES=4 MBAMD_MAX_LDS_SLOTS=4
  abl W2_slots3 $cfg MBAMD_WALK_WAVES=2 MBAMD_MAX_LDS_SLOTS=3
  abl W2_slots6 $cfg MBAMD_WALK_WAVES=2 MBAMD_MAX_LDS_SLOTS=6
  abl W1 $cfg MBAMD_WALK_WAVES=1
  abl W8_slots2 $cfg MBAMD_WALK_WAVES=8 MBAMD_MAX_LDS_SLOTS=2
done
