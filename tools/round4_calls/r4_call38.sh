#!/bin/bash
# Round 4, GPU call 38: fp64 general states -- the narrow levels at the top of the tree as one launch (chains walked per 64-pattern workgroup),
# matrix updates queued into one launch, unchanged weights / frequencies not re-sent.  Tests, evaluation times, dispatches.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_call38.log; : > $OUT
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "double or f64 or hazard or multi_partition" 2>&1 | tail -3 | tee -a $OUT
timeout 1200 python -m pytest tests/test_mrbayes_dropin.py -x -q -m gpu -k "double" 2>&1 | tail -3 | tee -a $OUT
echo "== all on" | tee -a $OUT; timeout 600 python tools/f64_bench.py c5 c3 c4 c2 2>&1 | grep config | tee -a $OUT
echo "== MBAMD_F64_NO_TAIL=1" | tee -a $OUT; MBAMD_F64_NO_TAIL=1 timeout 600 python tools/f64_bench.py c5 2>&1 | grep config | tee -a $OUT
echo "== MBAMD_F64_NO_MATRIX_QUEUE=1" | tee -a $OUT; MBAMD_F64_NO_MATRIX_QUEUE=1 timeout 600 python tools/f64_bench.py c5 c3 2>&1 | grep config | tee -a $OUT
cd /tmp
rm -rf /tmp/pf; F64_STEPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o x -- python $GRAFT_REPO_ROOT/tools/f64_bench.py c5 > /tmp/pf.log 2>&1
db=$(find /tmp/pf -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $db "" 22 | tee -a $OUT
