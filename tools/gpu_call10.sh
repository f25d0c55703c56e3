#!/bin/bash
# fp64 four-state tree walk: parity on the GPU, then time against the level kernels at C2 / C4 with a few slot budgets
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "double or other_state" > gpurun_out/pytest_f64.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_f64.log
timeout 900 python -m pytest tests/test_mrbayes_dropin.py -m gpu -x -q -k "double" > gpurun_out/pytest_f64_mb.log 2>&1; echo "pytest (binary) exit $?"; tail -3 gpurun_out/pytest_f64_mb.log
{
timeout 600 python tools/f64_time.py c2 c4
for sl in 2 3 6 9; do echo "== MBAMD_F64_WALK_SLOTS=$sl"; MBAMD_F64_WALK_SLOTS=$sl timeout 600 python tools/f64_time.py c4 | grep "fp64 engine"; done
} 2>&1 | tee gpurun_out/f64_time.txt
