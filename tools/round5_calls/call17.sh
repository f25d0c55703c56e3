#!/bin/bash
# round 5, call 17: k_path4 (root-ward paths of the 4-state walk in two passes) -- parity, partial-update time, fixed-topology MCMC rate, with / without
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "not codon and not protein" > gpurun_out/r5c17_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c17_pytest.log
tail -4 gpurun_out/r5c17_pytest.log
{
for v in "" 1; do
  echo "== MBAMD_NO_PATH4=$v"
  env ${v:+MBAMD_NO_PATH4=1} timeout 300 python tools/partial_time.py gtr 400
done
for v in "" 1; do
  echo "== MBAMD_NO_PATH4=$v: unmodified MrBayes, fixed topology, 500 x 20000, 12000 generations"
  env ${v:+MBAMD_NO_PATH4=1} timeout 300 python tools/mcmc_stats.py 500 20000 12000 dynamic fixed 2>&1 | grep -v 'Set\*\|Pars\|GetSite\|ScaleFactors'
done
} 2>&1 | tee gpurun_out/r5c17_path4.log
