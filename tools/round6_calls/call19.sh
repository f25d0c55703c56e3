#!/bin/bash
# call 19: forked paths (the lists of topology moves on k_path4 / k_path4_lnl) -- the new GPU test, the path tests again, the default
# mix with the device-parsimony binding with and without (MBAMD_NO_FORK_PATH=1), then call 18's sweep of waves per workgroup at C2.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c19; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "join or path or partial_update or one_launch" 2>&1 | tail -5 | tee gpurun_out/c19/tests.txt
timeout 1200 python tools/mcmc_ab.py pars mix 2000 32000 MBAMD_NO_FORK_PATH=1 2>&1 | tee gpurun_out/c19/ab_mix.txt
bash tools/round6_calls/call18.sh
