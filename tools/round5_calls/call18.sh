#!/bin/bash
# round 5, call 18: kernel timeline of fixed-topology generations with k_path4
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
st = mbdata.synthetic_states(500, 20000, 4, 7, 0.15, 0.0)
tr = mbtree.random_tree(500, 3, brlen=0.05)
open("/tmp/mc.nex", "w").write(refrun.mcmc_nexus(st, tr, 2000, beagle="dynamic", fixed_topology=True))
PY
rm -rf /tmp/prof_mc; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_mc -o mc -- $GRAFT_REPO_ROOT/oracle/_ref/mb_amd mc.nex > /tmp/mc.log 2>&1)
python - <<PY2 2>&1 | tee gpurun_out/r5c18_timeline.log
import sqlite3,glob,statistics
con=sqlite3.connect(glob.glob("/tmp/prof_mc/**/*.db",recursive=True)[0])
for pat in ("%path4%", "%walk4%", "%transition%", "%integrate%"):
    d=sorted(r[0] for r in con.execute("select duration from kernels where name like \"%s\"" % pat).fetchall())
    if d: print(pat, "launches",len(d),"median us",statistics.median(d)/1e3,"mean",sum(d)/len(d)/1e3, "p10",d[len(d)//10]/1e3,"p90",d[9*len(d)//10]/1e3)
PY2
python tools/rocpd_timeline.py $(find /tmp/prof_mc -name "*.db" | head -1) 30 2>&1 | tee -a gpurun_out/r5c18_timeline.log
