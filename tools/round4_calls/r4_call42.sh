#!/bin/bash
# Round 4, GPU call 42: fp64 codon, narrow levels with the two children contracted side by side (k64_partials_mfma_split): tests, time, dispatches
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_call42.log; : > $OUT
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "double or f64 or hazard or multi_partition" 2>&1 | tail -3 | tee -a $OUT
timeout 1200 python -m pytest tests/test_mrbayes_dropin.py -x -q -m gpu -k "double" 2>&1 | tail -3 | tee -a $OUT
echo "== default (split up to 256 workgroups)" | tee -a $OUT; timeout 600 python tools/f64_bench.py c5 2>&1 | grep config | tee -a $OUT
echo "== MBAMD_F64_SPLIT_MAX=512" | tee -a $OUT; MBAMD_F64_SPLIT_MAX=512 timeout 600 python tools/f64_bench.py c5 2>&1 | grep config | tee -a $OUT
echo "== MBAMD_F64_NO_SPLIT=1" | tee -a $OUT; MBAMD_F64_NO_SPLIT=1 timeout 600 python tools/f64_bench.py c5 2>&1 | grep config | tee -a $OUT
cd /tmp
rm -rf /tmp/pf; MBAMD_F64_SPLIT_MAX=512 F64_STEPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o x -- python $GRAFT_REPO_ROOT/tools/f64_bench.py c5 > /tmp/pf.log 2>&1
db=$(find /tmp/pf -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $db "k64_partials" 17 | tee -a $OUT
