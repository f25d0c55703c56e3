#!/bin/bash
# Round 4, GPU call 26 (experiment build): clock stamps inside k_walk4_t for root-ward path programs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
# (the stamps were a temporary patch of k_walk4_t and its reader a throw-away script: both gone; what they showed is in profiles/r04_mcmc_fixed_topology.txt)
MBAMD_WALK_TRACE=1 MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_trace4.so timeout 600 python tools/trace_walk4.py > gpurun_out/r4_call26.log 2>&1
tail -5 gpurun_out/r4_call26.log
