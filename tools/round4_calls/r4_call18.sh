#!/bin/bash
# Round 4, GPU call 18: transition-matrix tables written front to back from an LDS tile; the whole GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call18.log; : > $OUT
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee -a $OUT
for c in c5 c3; do TIMELINE=3 bash tools/prof_one.sh $c > /dev/null 2>&1; head -8 gpurun_out/prof_${c}_summary.txt | cut -c1-170 | tee -a $OUT; head -3 gpurun_out/prof_${c}_timeline.txt | tee -a $OUT; grep '^{' gpurun_out/prof_$c.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['value'], d['ms_per_step'], r['frac'])" | tee -a $OUT; done
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_engine_gpu.py --maxfail=10 2>&1 | tail -4 | tee -a $OUT
