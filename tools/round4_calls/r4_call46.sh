#!/bin/bash
# Round 4, GPU call 46: the matrices parked with 16-byte loads / LDS writes: tests, time, and the stamp build (where a wave's time goes)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_call46.log; : > $OUT
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "double or f64 or hazard or multi_partition" 2>&1 | tail -3 | tee -a $OUT
timeout 600 python tools/f64_bench.py c5 c3 2>&1 | grep config | tee -a $OUT
MBAMD_LIBRARY=build_x/libhmsbeagle_stamps.so F64_STEPS=20 timeout 200 python tools/f64_bench.py c5 2>&1 | tail -3 | tee -a $OUT
