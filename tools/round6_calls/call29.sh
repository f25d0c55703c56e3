#!/bin/bash
# call 29: no stream flag behind an integration whose block sums are awaited -- full GPU suite, the two chains (the previous commit's library beside it)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c29; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c29/gputests.txt
timeout 900 python tools/mcmc_ab.py amd fixed 2000 42000 MBAMD_NO_SUM_POLL=1 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite\|Pars\|pars' | tee gpurun_out/c29/fixed.txt
timeout 1200 python tools/mcmc_ab.py pars mix 2000 32000 MBAMD_NO_SUM_POLL=1 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite' | tee gpurun_out/c29/mix.txt
