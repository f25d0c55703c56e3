#!/bin/bash
# call 46: a walk program reaches its device buffer through a copy kernel of ours (MBAMD_NO_RING_COPY=1: hipMemcpyAsync, as before) -- 4-state GPU
# tests, the two chains with / without, the kernels of 2 000 fixed-topology generations
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c46; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/c46/gputests.txt
timeout 900 python tools/mcmc_ab.py amd fixed 2000 42000 MBAMD_NO_RING_COPY=1 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite\|Pars\|pars' | tee gpurun_out/c46/fixed.txt
timeout 1200 python tools/mcmc_ab.py pars mix 2000 32000 MBAMD_NO_RING_COPY=1 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite' | tee gpurun_out/c46/mix.txt
timeout 600 bash tools/prof_mcmc.sh 2>&1 | head -30 | tee gpurun_out/c46/timeline.txt
