#!/bin/bash
# call 25: exponents stored only by SCALE_WRITE entries (4-state walk and path kernels) -- full GPU suite, a full-tree evaluation
# under both rescaling schemes at the four BASELINE shapes (the general-state kernels still store to the scratch row), the two chains.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c25; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c25/gputests.txt
{ timeout 600 python tools/scale_read_time.py bench_c2 200; timeout 900 python tools/scale_read_time.py bench_c4 60;
  timeout 600 python tools/scale_read_time.py bench_c3 100; timeout 600 python tools/scale_read_time.py bench_c5 100; } 2>&1 | tee gpurun_out/c25/scale_read.txt
timeout 900 python tools/mcmc_ab.py amd fixed 2000 42000 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite\|Pars\|pars' | tee gpurun_out/c25/fixed.txt
timeout 1200 python tools/mcmc_ab.py pars mix 2000 32000 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite' | tee gpurun_out/c25/mix.txt
