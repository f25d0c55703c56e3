#!/bin/bash
# round 5, call 25: noise of the two-point generations/s measurement (fixed topology), and the rate with / without k_path4 and its fused matrices
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 300 python tools/mcmc_walls.py 2000 12000 42000
timeout 300 python tools/mcmc_walls.py 2000 42000 MBAMD_NO_PATH_MATRICES=1
timeout 300 python tools/mcmc_walls.py 2000 42000 MBAMD_NO_PATH4=1
} 2>&1 | tee gpurun_out/r5c25.log
