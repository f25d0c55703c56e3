#!/bin/bash
# round 5, call 40: the kernels of the codon M3 chain (100 x 5 000, fixed topology, every binding) and of the protein chain (200 x 10 000):
# rocprofv3 summary + a stretch of the timeline + engine statistics
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
for case, kind, ns, ngen in (("bench_c5", "m3", 61, 1500), ("bench_c3", "wag", 20, 3000)):
    with open(os.path.join(bench.GOLD, case + ".json")) as fh:
        g = json.load(fh)
    s = g["synthetic"]
    st = mbdata.synthetic_states(s["ntaxa"], s["nsites"], ns, s["seed"], s["p_mut"], s["p_gap"])
    tr = mbtree.parse_newick(g["newick"])
    open("/tmp/%s.nex" % kind, "w").write(refrun.model_nexus(kind, st, tr, ngen=ngen, beagle="dynamic", fixed_topology=True))
PY
for kind in m3 wag; do
  rm -rf /tmp/prof_$kind; (cd /tmp && MBAMD_STATS=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$kind -o s -- $GRAFT_REPO_ROOT/oracle/_ref/mb_amd_full $kind.nex > /tmp/$kind.log 2>&1)
  db=$(find /tmp/prof_$kind -name "*.db" | head -1)
  echo "== $kind"; grep '\[mbamd\]' /tmp/$kind.log | cut -c1-120
  python tools/rocpd_summary.py $db | cut -c1-170 | head -16
  python - $db <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end, queue_id from kernels order by start").fetchall()
i0 = len(rows) * 2 // 3
t0 = rows[i0][1]; prev_end = t0
for r in rows[i0:i0 + 44]:
    print("%9.1f  dur %6.1f  gap %6.1f  q %s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3, r[3], r[0][:56]))
    prev_end = max(prev_end, r[2])
PY
done 2>&1 | tee gpurun_out/r5c40.log
