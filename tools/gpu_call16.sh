#!/bin/bash
# fp64 matrix-core level kernel for 16..64 states: parity, then device time against round 2's vector-ALU kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "double or multi_partition" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_mrbayes_dropin.py -m gpu -x -q -k "double" 2>&1 | tail -2
timeout 600 python tools/f64_time.py c3 c5 2>&1 | tee gpurun_out/f64_time_general.txt
bash tools/prof_tool.sh f64_gen $PWD/tools/f64_time.py c3 c5 > /dev/null 2>&1
grep -E "k64_" gpurun_out/prof_f64_gen_summary.txt | cut -c1-175
