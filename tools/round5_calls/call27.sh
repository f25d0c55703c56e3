#!/bin/bash
# round 5, call 27: where a default-move-mix generation goes with the device-parsimony binding (MBAMD_STATS=1, 1 000 and 6 000 generations)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/mcmc_pars_stats.py 2>&1 | tee gpurun_out/r5c27.log | tail -80
