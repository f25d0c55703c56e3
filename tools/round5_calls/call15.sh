#!/bin/bash
# round 5, call 15: after the refactoring (k_walkg default, k_walkg2 / row split behind MBAMD_WALKG_PAIR=1, k_walkg_s gone): the engine's GPU tests, A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu > gpurun_out/r5c15_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c15_pytest.log
tail -5 gpurun_out/r5c15_pytest.log
for c in c5 c3; do
  timeout 120 python tools/ablate_walkg.py $c
  MBAMD_WALKG_PAIR=1 timeout 120 python tools/ablate_walkg.py $c
done 2>&1 | tee gpurun_out/r5c15_ab.log
