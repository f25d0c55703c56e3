#!/bin/bash
# Round 4, GPU call 43: fp64 contraction kernel -- a tip child's column from the parked matrix in LDS instead of a gather from global memory
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_call43.log; : > $OUT
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "double or f64 or hazard or multi_partition" 2>&1 | tail -3 | tee -a $OUT
timeout 1200 python -m pytest tests/test_mrbayes_dropin.py -x -q -m gpu -k "double" 2>&1 | tail -3 | tee -a $OUT
timeout 600 python tools/f64_bench.py c5 c3 2>&1 | grep config | tee -a $OUT
cd /tmp
rm -rf /tmp/pf; F64_STEPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o x -- python $GRAFT_REPO_ROOT/tools/f64_bench.py c5 > /tmp/pf.log 2>&1
db=$(find /tmp/pf -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $db "k64_partials" 17 | tee -a $OUT
