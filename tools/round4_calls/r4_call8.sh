#!/bin/bash
# Round 4, GPU call 8: the GPU suite after the final-pass / std-marker fixes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
