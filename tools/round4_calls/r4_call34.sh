#!/bin/bash
# Round 4, GPU call 34: the fp64 codon evaluation dispatch by dispatch (where the level-synchronous time goes)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_call34.log; : > $OUT
cd /tmp
rm -rf /tmp/pf; F64_STEPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o x -- python $GRAFT_REPO_ROOT/tools/f64_bench.py c5 > /tmp/pf.log 2>&1
db=$(find /tmp/pf -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $db "" 60 | tee -a $OUT
python - "$db" <<'PY' | tee -a $OUT
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
print([r[1] for r in con.execute("pragma table_info(kernels)")])
PY
