#!/bin/bash
# Round 4, GPU call 41: the LDS contraction kernel at 20 states x 4 categories again, now that all categories' matrices are parked in one batch
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_call41.log; : > $OUT
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "double or f64" 2>&1 | tail -2 | tee -a $OUT
MBAMD_F64_MFMA_LDS_ALL=1 timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "test_double_precision and (avian or aa_wag or bench_c3)" 2>&1 | tail -2 | tee -a $OUT
echo "== default" | tee -a $OUT; timeout 600 python tools/f64_bench.py c3 c5 2>&1 | grep config | tee -a $OUT
echo "== MBAMD_F64_MFMA_LDS_ALL=1" | tee -a $OUT; MBAMD_F64_MFMA_LDS_ALL=1 timeout 600 python tools/f64_bench.py c3 2>&1 | grep config | tee -a $OUT
echo "== MBAMD_F64_MFMA_LDS_ALL=1 MBAMD_F64_MFMA_4WAVES=1" | tee -a $OUT; MBAMD_F64_MFMA_4WAVES=1 MBAMD_F64_MFMA_LDS_ALL=1 timeout 600 python tools/f64_bench.py c3 2>&1 | grep config | tee -a $OUT
