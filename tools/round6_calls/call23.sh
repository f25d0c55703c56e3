#!/bin/bash
# call 23: stored exponents requested six entries ahead (ring of eight landing areas) -- full GPU suite, a full-tree evaluation
# under both rescaling schemes (call 22 measured 0.1226 / 0.1659 ms at DNA 500 x 20 000 before), the two chains.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c23; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c23/gputests.txt
{ timeout 600 python tools/scale_read_time.py bench_c2 200; timeout 900 python tools/scale_read_time.py bench_c4 60; } 2>&1 | tee gpurun_out/c23/scale_read.txt
timeout 900 python tools/mcmc_ab.py amd fixed 2000 42000 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite\|Pars\|pars' | tee gpurun_out/c23/fixed.txt
timeout 1200 python tools/mcmc_ab.py pars mix 2000 32000 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite' | tee gpurun_out/c23/mix.txt
