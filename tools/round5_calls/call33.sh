#!/bin/bash
# round 5, call 33: the kernels of a default-move-mix chain with the device-parsimony binding (3 000 generations under rocprofv3)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
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
rm -rf /tmp/prof_mcp; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_mcp -o mc -- $GRAFT_REPO_ROOT/oracle/_ref/mb_amd_pars mcp.nex > /tmp/mcp.log 2>&1)
db=$(find /tmp/prof_mcp -name "*.db" | head -1)
python tools/rocpd_summary.py $db | cut -c1-190 | head -14 | tee gpurun_out/r5c33.log
python - $db <<'PY' | tee -a gpurun_out/r5c33.log
import sqlite3, sys, statistics
con = sqlite3.connect(sys.argv[1])
for pat in ("k_pars_walk", "k_pars_score"):
    d = sorted(r[0] for r in con.execute('select duration from kernels where name like "%%%s%%"' % pat).fetchall())
    if d: print(pat, "launches", len(d), "median us %.1f mean %.1f p10 %.1f p90 %.1f max %.1f" % (statistics.median(d)/1e3, sum(d)/len(d)/1e3, d[len(d)//10]/1e3, d[9*len(d)//10]/1e3, d[-1]/1e3))
PY
