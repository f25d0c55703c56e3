#!/bin/bash
# call 49: the final tree once more: GPU suite and smoke()
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/c49; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/c49/gputests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -c 300 | tee gpurun_out/c49/smoke.txt
