#!/bin/bash
# call 26: general-state entries without SCALE_WRITE rotate over 32 scratch exponent rows -- full GPU suite, a full-tree evaluation
# under both rescaling schemes at protein 200 x 10 000 and codon 100 x 5 000 (call 25: 0.2126 / 0.2745 and 0.1580 / 0.1579 ms)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c26; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c26/gputests.txt
{ timeout 600 python tools/scale_read_time.py bench_c3 100; timeout 600 python tools/scale_read_time.py bench_c5 100; } 2>&1 | tee gpurun_out/c26/scale_read.txt
# (added after a transient refusal) the matrix kernel launched by beagleUpdateTransitionMatrices itself (four states): A/B on the two chains
timeout 900 python tools/mcmc_ab.py amd fixed 2000 42000 MBAMD_EAGER_MATRICES=1 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite\|Pars\|pars' | tee gpurun_out/c26/fixed.txt
timeout 1200 python tools/mcmc_ab.py pars mix 2000 32000 MBAMD_EAGER_MATRICES=1 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite' | tee gpurun_out/c26/mix.txt
