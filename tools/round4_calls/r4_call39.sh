#!/bin/bash
# Round 4, GPU call 39: fp64 general states -- both matrices parked in one batch of loads, the first group of partials in flight meanwhile;
# bit-equality test of all new paths against the plain level kernel; the tail launch on / off.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_call39.log; : > $OUT
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "double or f64 or hazard or multi_partition" 2>&1 | tail -3 | tee -a $OUT
timeout 1200 python -m pytest tests/test_mrbayes_dropin.py -x -q -m gpu -k "double" 2>&1 | tail -3 | tee -a $OUT
echo "== all on" | tee -a $OUT; timeout 600 python tools/f64_bench.py c5 c3 2>&1 | grep config | tee -a $OUT
echo "== MBAMD_F64_NO_TAIL=1" | tee -a $OUT; MBAMD_F64_NO_TAIL=1 timeout 600 python tools/f64_bench.py c5 2>&1 | grep config | tee -a $OUT
cd /tmp
rm -rf /tmp/pf; MBAMD_F64_NO_TAIL=1 F64_STEPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o x -- python $GRAFT_REPO_ROOT/tools/f64_bench.py c5 > /tmp/pf.log 2>&1
db=$(find /tmp/pf -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $db "k64_partials" 18 | tee -a $OUT
rm -rf /tmp/pf; F64_STEPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o x -- python $GRAFT_REPO_ROOT/tools/f64_bench.py c5 > /tmp/pf.log 2>&1
db=$(find /tmp/pf -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $db "tail" 2 | tee -a $OUT
