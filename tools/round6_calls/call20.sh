#!/bin/bash
# call 20: the block sums as their own completion signal (MBAMD_NO_SUM_POLL=1 for A/B) and three waves per workgroup at DNA 500 x
# 20 000 -- full GPU suite, the two chains with and without, the four workloads' step times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c20; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c20/gputests.txt
timeout 900 python tools/mcmc_ab.py amd fixed 2000 42000 MBAMD_NO_SUM_POLL=1 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite\|Pars' | tee gpurun_out/c20/ab_fixed.txt
timeout 1200 python tools/mcmc_ab.py pars mix 2000 32000 MBAMD_NO_SUM_POLL=1 2>&1 | grep -v 'beagleSet\|ScaleFactors\|GetSite' | tee gpurun_out/c20/ab_mix.txt
for v in "" 1; do
  echo "== MBAMD_NO_SUM_POLL=$v"
  env ${v:+MBAMD_NO_SUM_POLL=1} timeout 900 python bench.py --config c2 --steps 200 --no-cpu-baseline --no-mpi --no-arith --no-mcmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: v for k, v in d['summary'].items() if k in ('c2','c3','c4','c5')})"
done 2>&1 | tee gpurun_out/c20/steps.txt
