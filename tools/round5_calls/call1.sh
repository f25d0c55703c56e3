#!/bin/bash
# round 5, call 1: the row-split 61-state walk -- parity first, then A/B against round 4's kernel (build_x/libhmsbeagle_r4.so) on this box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu > gpurun_out/r5c1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c1_pytest.log
tail -5 gpurun_out/r5c1_pytest.log
for i in 1 2; do
  timeout 120 python tools/ablate_walkg.py c5
  MBAMD_LIBRARY=build_x/libhmsbeagle_r4.so timeout 120 python tools/ablate_walkg.py c5
done 2>&1 | tee gpurun_out/r5c1_ab.log
timeout 120 python tools/ablate_walkg.py c3 2>&1 | tee -a gpurun_out/r5c1_ab.log
