#!/bin/bash
# Round 4, GPU call 5: k_walkg with idle operand loads collapsed to one request (tips' later chunks, no-op entries) vs round 3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call5.log; : > $OUT
say() { echo "$@" | tee -a $OUT; }
abl() { local label=$1 cfg=$2; shift 2; env "$@" timeout 300 python tools/ablate_walkg.py $cfg 2>/tmp/abl.err | tail -1 | sed "s/product/$label/" | tee -a $OUT; }
export MBAMD_WALKG_SHARED=0
timeout 1500 python -m pytest tests/test_engine_gpu.py -q -m gpu --maxfail=10 > gpurun_out/r4_pytest_gpu5.log 2>&1; say "engine gpu tests (k_walkg) exit $?"; tail -4 gpurun_out/r4_pytest_gpu5.log | tee -a $OUT
for rep in 1 2; do for cfg in c5 c3; do
  abl r4_walkg $cfg X=1
  abl r3_walkg $cfg MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_r3.so
done; done
