#!/bin/bash
# Round 4, GPU call 51: k64_partials_chain (a root-ward path in one launch, the running result in registers): tests, then the double-precision chains of call 49
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call51.log; : > $OUT
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "double or f64 or hazard or multi_partition" 2>&1 | tail -5 | tee -a $OUT
timeout 1200 python -m pytest tests/test_mrbayes_dropin.py -x -q -m gpu -k "double" 2>&1 | tail -3 | tee -a $OUT
bash tools/round4_calls/r4_call49.sh 2>&1 | tail -4 | tee -a $OUT
