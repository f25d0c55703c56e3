#!/bin/bash
# Round 4, GPU call 16: fp64 after the wide integration; host-side statistics of a fixed-topology MCMC run (where a generation's time goes)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call16.log; : > $OUT
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "double or f64 or hazard or multi_partition" 2>&1 | tail -3 | tee -a $OUT
timeout 600 python tools/f64_bench.py c5 c3 2>&1 | tail -2 | tee -a $OUT
echo "== fixed topology, 500 x 20000, dynamic scaling, MBAMD_STATS" | tee -a $OUT
timeout 900 python tools/mcmc_stats.py 500 20000 12000 dynamic fixed 2>&1 | tail -40 | tee -a $OUT
echo "== the same with ROC_ACTIVE_WAIT_TIMEOUT=1000" | tee -a $OUT
ROC_ACTIVE_WAIT_TIMEOUT=1000 timeout 900 python tools/mcmc_stats.py 500 20000 12000 dynamic fixed 2>&1 | grep -i "wall\|wait" | tee -a $OUT
