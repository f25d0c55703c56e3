#!/bin/bash
# round 5, call 9: k_walkg2 -- parity, then kernel time at C5 / C3 against round 4's kernel, wave durations
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu > gpurun_out/r5c9_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c9_pytest.log
tail -12 gpurun_out/r5c9_pytest.log
for c in c5 c3; do
  timeout 120 python tools/ablate_walkg.py $c
  MBAMD_LIBRARY=build_x/libhmsbeagle_r4.so timeout 120 python tools/ablate_walkg.py $c
done 2>&1 | tee gpurun_out/r5c9_ab.log
for c in c5 c3; do
  echo "== w2 $c"
  MBAMD_LIBRARY=build_x/libhmsbeagle_w2.so timeout 120 python tools/trace_walkg.py $c 2>&1 | grep -v '^(\|^wave\|^bench'
done 2>&1 | tee gpurun_out/r5c9_trace.log
