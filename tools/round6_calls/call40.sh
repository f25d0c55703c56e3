#!/bin/bash
# call 40: the default bench line once more on the final code (with roofline.issue_bound_frac from the PMC passes committed in profiles/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench exit $?"; tail -c 600 gpurun_out/bench_final.json
