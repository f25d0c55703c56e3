#!/bin/bash
# usage: prof_tool.sh <tag> <python tool and its arguments...>  -> gpurun_out/prof_<tag>_summary.txt  (rocprofv3 --kernel-trace --stats)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=$1; shift
rm -rf /tmp/prof_$tag
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag -- python "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1)
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_summary.py $db | tee gpurun_out/prof_${tag}_summary.txt | cut -c1-200
