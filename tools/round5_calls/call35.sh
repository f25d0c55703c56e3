#!/bin/bash
# round 5, call 35: a parsimony move on the timeline (default mix, device-parsimony binding, 3 000 generations under rocprofv3)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "parsimony" 2>&1 | tail -2
python - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
with open(os.path.join(bench.GOLD, bench.CONFIGS["c2"][0] + ".json")) as fh:
    gold = json.load(fh)
sy = gold["synthetic"]
st = mbdata.synthetic_states(sy["ntaxa"], sy["nsites"], 4, sy["seed"], sy["p_mut"], sy["p_gap"])
tr = mbtree.parse_newick(gold["newick"])
open("/tmp/mcp.nex", "w").write(refrun.mcmc_nexus(st, tr, 3000, beagle="dynamic"))
PY
rm -rf /tmp/prof_mcp; (cd /tmp && MBAMD_VERBOSE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_mcp -o mc -- $GRAFT_REPO_ROOT/oracle/_ref/mb_amd_pars mcp.nex > /tmp/mcp.log 2>&1)
db=$(find /tmp/prof_mcp -name "*.db" | head -1)
{
python tools/rocpd_summary.py $db | cut -c1-190 | head -12
grep "parsimony program" /tmp/mcp.log | head -8 | cut -c1-300
python - $db <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_pars_walk" in r[0]]
for i0 in idx[5:8]:
    t0 = rows[max(0, i0 - 3)][1]
    prev_end = t0
    for r in rows[max(0, i0 - 3): i0 + 8]:
        print("%9.1f  dur %7.1f  gap %6.1f  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3, r[0][:60]))
        prev_end = r[2]
    print("--")
PY
} 2>&1 | tee gpurun_out/r5c35.log
