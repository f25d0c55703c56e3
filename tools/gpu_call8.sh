#!/bin/bash
# 2- and 8-state instantiations of the tree-walk kernel: the whole GPU suite, then walk against level kernels at a fixed shape
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
{ for s in 2 8 16; do timeout 300 python tools/states_time.py $s 200 10000; done; } 2>&1 | tee gpurun_out/states_time.log
