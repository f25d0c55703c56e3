#!/bin/bash
# Round 4, GPU call 29 (experiment build): waiting for the result by polling an event recorded behind the integration kernel (hipEventQuery)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call29.log; : > $OUT
L=$PWD/build_x/libhmsbeagle_pollev.so
run() { echo "== $*" | tee -a $OUT; env "$@" timeout 900 python tools/mcmc_stats.py 500 20000 12000 dynamic fixed 2>&1 | grep -i "^wall\|waiting for\|Analysis completed\|Analysis used" | tee -a $OUT; }
run LD_PRELOAD=$L X=1
run LD_PRELOAD=$L MBAMD_POLL_EVENT=1
run LD_PRELOAD=$L MBAMD_NO_POLL=1
run LD_PRELOAD=$L MBAMD_POLL_EVENT=1
