#!/bin/bash
# Round 4, GPU call 19: where do a path walk's 20 us go?  inline programs (kernel arguments) against a device buffer, kernarg placement
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call19.log; : > $OUT
run() { echo "== $*" | tee -a $OUT; env "$@" timeout 900 python tools/mcmc_stats.py 500 20000 12000 dynamic fixed 2>&1 | grep -i "^wall\|waiting for\|Analysis completed\|UpdatePartials\|plan build" | tee -a $OUT; }
run X=1
run MBAMD_NO_INLINE_PROGRAMS=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1 MBAMD_NO_INLINE_PROGRAMS=1
