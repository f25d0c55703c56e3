#!/bin/bash
# Round 4, GPU call 22: the up-front burst of short 4-state walk programs (root-ward paths) -- engine tests, MCMC statistics with and without
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call22.log; : > $OUT
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee -a $OUT
run() { echo "== $*" | tee -a $OUT; env "$@" timeout 900 python tools/mcmc_stats.py 500 20000 12000 dynamic fixed 2>&1 | grep -i "^wall\|waiting for\|Analysis completed\|Analysis used" | tee -a $OUT; }
run X=1
run MBAMD_WALK_NO_WARM=1
run X=2
run MBAMD_WALK_NO_WARM=1
bash tools/prof_mcmc.sh 2>&1 | head -1 | tee -a $OUT

