#!/bin/bash
# usage: prof_partial.sh <gtr|wag|m3> [env assignments...]  -> kernel-trace summary of tools/partial_time.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
m=${1:-wag}; shift
tag=${TAG:-$m}
rm -rf /tmp/profp_$tag
(cd /tmp && env "$@" MBAMD_STATS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/profp_$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/partial_time.py $m 300 > $GRAFT_REPO_ROOT/gpurun_out/profp_$tag.log 2>&1)
db=$(find /tmp/profp_$tag -name "*.db" | head -1)
python tools/rocpd_summary.py $db | tee gpurun_out/profp_${tag}_summary.txt | cut -c1-190
grep -v "^W2\|rocprof" gpurun_out/profp_$tag.log | tail -25
