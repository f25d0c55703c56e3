#!/bin/bash
# usage: prof_one.sh <config> [extra env assignments...]   -> gpurun_out/prof_<config>_summary.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
c=${1:-c2}
rm -rf /tmp/prof_$c
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o $c -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-mcmc --no-arith > $GRAFT_REPO_ROOT/gpurun_out/prof_$c.log 2>&1)
db=$(find /tmp/prof_$c -name "*.db" | head -1)
python tools/rocpd_summary.py $db | tee gpurun_out/prof_${c}_summary.txt | cut -c1-200
grep '^{' gpurun_out/prof_$c.log | cut -c1-400
python tools/rocpd_timeline.py $db ${TIMELINE:-0} > gpurun_out/prof_${c}_timeline.txt 2>/dev/null || true
