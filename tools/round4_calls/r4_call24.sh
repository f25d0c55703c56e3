#!/bin/bash
# Round 4, GPU call 24 (experiment): waiting for the result by polling a word the stream writes behind the integration kernel, against hipStreamSynchronize
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call24.log; : > $OUT
L=$PWD/build_x/libhmsbeagle_poll.so
cp oracle/_ref/mb_amd /tmp/mb_poll
run() { echo "== $*" | tee -a $OUT; env "$@" timeout 900 python tools/mcmc_stats.py 500 20000 12000 dynamic fixed 2>&1 | grep -i "^wall\|waiting for\|Analysis completed\|Analysis used" | tee -a $OUT; }
run LD_PRELOAD=$L X=1
run LD_PRELOAD=$L MBAMD_POLL_RESULT=1
run LD_PRELOAD=$L X=2
run LD_PRELOAD=$L MBAMD_POLL_RESULT=1
