#!/bin/bash
# Round 4, GPU call 40: the fp64 tests after the clean-up (tail launch removed) and the four workloads' evaluation times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_call40.log; : > $OUT
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "double or f64 or hazard or multi_partition" 2>&1 | tail -3 | tee -a $OUT
timeout 1200 python -m pytest tests/test_mrbayes_dropin.py -x -q -m gpu -k "double" 2>&1 | tail -3 | tee -a $OUT
timeout 600 python tools/f64_bench.py c5 c3 c4 c2 2>&1 | grep config | tee -a $OUT
