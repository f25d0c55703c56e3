#!/bin/bash
# call 28: k_path4 / k_path4_lnl with a chunk's matrices staged in LDS (three vector loads instead of twelve scalar round trips) -- GPU path
# tests, the time of a path + log-likelihood launch (call 27: 21.70 us per launch averaged with 6 tree walks in 305), the two chains
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c28; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c28/gputests.txt
{ timeout 600 python tools/path_time.py bench_c2 300; MBAMD_NO_FUSE_PATH=1 timeout 600 python tools/path_time.py bench_c2 300;
  MBAMD_LIBRARY=$PWD/build_x/libhmsbeagle_prev.so timeout 600 python tools/path_time.py bench_c2 300; } 2>&1 | tee gpurun_out/c28/path_time.txt
timeout 900 python tools/mcmc_ab.py amd fixed 2000 42000 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite\|Pars\|pars' | tee gpurun_out/c28/fixed.txt
timeout 1200 python tools/mcmc_ab.py pars mix 2000 32000 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite' | tee gpurun_out/c28/mix.txt
