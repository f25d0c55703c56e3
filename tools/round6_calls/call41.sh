#!/bin/bash
# call 41: k_walkg<5..8> and k_partials_tips<0,2> out of scratch memory -- the full GPU suite (standard data with 5..8 states runs those)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c41; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/c41/gputests.txt
