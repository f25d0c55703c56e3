#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_round.sh tests
echo "== tip-ahead parity"; MBAMD_WALK_TIP_AHEAD=8 MBAMD_WALK_TIP_FROM=4 timeout 900 python -m pytest tests/test_engine_gpu.py -x -q 2>&1 | tail -3 | tee gpurun_out/pytest_tipahead.log
bash tools/exp_walk4.sh
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench default exit $?"
for c in c4 c2 c3 c5; do TIMELINE=16 bash tools/prof_one.sh $c > /dev/null 2>&1; head -8 gpurun_out/prof_${c}_summary.txt | cut -c1-160; done
