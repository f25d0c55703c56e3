#!/bin/bash
# round 5, call 19: whole GPU suite with k_path4 (program through LDS), fp64 launch bounds, std binding, bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5c19_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c19_pytest.log
tail -6 gpurun_out/r5c19_pytest.log
for v in "" 1; do
  echo "== MBAMD_NO_PATH4=$v: unmodified MrBayes, fixed topology, 500 x 20000, 12000 generations"
  env ${v:+MBAMD_NO_PATH4=1} timeout 300 python tools/mcmc_stats.py 500 20000 12000 dynamic fixed 2>&1 | grep 'wall\|plan build\|waiting\|UpdatePartials'
done 2>&1 | tee gpurun_out/r5c19_path4.log
