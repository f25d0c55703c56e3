#!/bin/bash
# round 5, call 38: standard data -- the kernels of one evaluation (five transition-matrix classes, five streams) on the timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from tests import std_cases
kw = dict(std_cases.BIG, ngen=2000)
nex = std_cases.synthetic_nexus(beagle="dynamic", **kw).replace(" startvals tau=t V=t;\n", "")
open("/tmp/std.nex", "w").write(nex)
PY
rm -rf /tmp/prof_std; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_std -o s -- $GRAFT_REPO_ROOT/oracle/_ref/mb_amd_full std.nex > /tmp/std.log 2>&1)
db=$(find /tmp/prof_std -name "*.db" | head -1)
{
python tools/rocpd_summary.py $db | cut -c1-170 | head -14
python - $db <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
print(cols)
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = con.execute("select name, start, end%s from kernels order by start" % ((", " + q) if q else "")).fetchall()
i0 = len(rows) // 2
t0 = rows[i0][1]
prev_end = t0
for r in rows[i0:i0 + 60]:
    print("%9.1f  dur %6.1f  gap %6.1f  q %s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3, r[3] if q else "-", r[0][:50]))
    prev_end = max(prev_end, r[2])
PY
} 2>&1 | tee gpurun_out/r5c38.log
