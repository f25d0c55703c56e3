#!/bin/bash
# Round 4, GPU call 52: the double-precision protein chain under rocprofv3 --kernel-trace: how long the chain launches take
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r4_call52.log; : > $OUT
python - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
g = json.load(open(os.path.join(bench.GOLD, "bench_c3.json")))
s = g["synthetic"]
st = mbdata.synthetic_states(s["ntaxa"], s["nsites"], 20, s["seed"], s["p_mut"], s["p_gap"])
tr = mbtree.parse_newick(g["newick"])
open("/tmp/c3d.nex", "w").write(refrun.model_nexus("wag", st, tr, ngen=400, beagle="dynamic", fixed_topology=True, precision="double", fname="/tmp/c3d"))
print(refrun.REF_MB_AMD)
PY
cd /tmp
rm -rf /tmp/pf; LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/mrbayes_amd:${LD_LIBRARY_PATH:-} timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o x -- $GRAFT_REPO_ROOT/oracle/_ref/mb_amd /tmp/c3d.nex > /tmp/pf.log 2>&1
tail -3 /tmp/pf.log
db=$(find /tmp/pf -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db | cut -c1-180 | head -12 | tee -a $OUT
python - "$db" <<'PY' | tee -a $OUT
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select duration, grid_y from kernels where name like '%k64_partials_chain%'").fetchall()
print("chain launches:", len(rows))
import statistics
if rows:
    print("duration us: median %.1f  mean %.1f  min %.1f  max %.1f" % (statistics.median(r[0] for r in rows) / 1e3, statistics.mean(r[0] for r in rows) / 1e3, min(r[0] for r in rows) / 1e3, max(r[0] for r in rows) / 1e3))
PY
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $db "" 30 | tee -a $OUT
