#!/bin/bash
# Round 4, GPU call 10: the fp64 four-state walk with its operands fetched one entry ahead
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call10.log; : > $OUT
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "double or f64 or hazard or multi_partition" 2>&1 | tail -5 | tee -a $OUT
timeout 600 python tools/f64_bench.py c4 c2 c1 2>&1 | tail -4 | tee -a $OUT
for s in 2 3 6 8 12; do echo "slots $s" | tee -a $OUT; MBAMD_F64_WALK_SLOTS=$s timeout 600 python tools/f64_bench.py c4 2>&1 | tail -1 | tee -a $OUT; done
