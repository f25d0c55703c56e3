#!/bin/bash
# Round 4, GPU call 20: fp64 level kernel on the matrix cores with two 16-pattern tiles per wave (codon parts)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call20.log; : > $OUT
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "double or f64 or hazard or multi_partition" 2>&1 | tail -3 | tee -a $OUT
timeout 1200 python -m pytest tests/test_mrbayes_dropin.py -x -q -m gpu -k "double" 2>&1 | tail -3 | tee -a $OUT
timeout 600 python tools/f64_bench.py c5 c3 2>&1 | tail -2 | tee -a $OUT
MBAMD_F64_ONE_TILE=1 timeout 600 python tools/f64_bench.py c5 2>&1 | tail -1 | sed 's/^/one tile: /' | tee -a $OUT
cd /tmp
rm -rf /tmp/pf; F64_STEPS=10 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o x -- python $GRAFT_REPO_ROOT/tools/f64_bench.py c5 > /tmp/pf.log 2>&1
db=$(find /tmp/pf -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db | grep -i "k64\|Name" | cut -c1-200 | tee -a $GRAFT_REPO_ROOT/$OUT
