#!/bin/bash
# Round 4, GPU call 50: host-side statistics (MBAMD_STATS=1) of the double-precision chains of call 49
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call50.log; : > $OUT
timeout 900 python - <<'PY' 2>&1 | tee -a $OUT
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
for case, kind, nst in (("bench_c5", "m3", 61), ("bench_c3", "wag", 20)):
    g = json.load(open(os.path.join(bench.GOLD, case + ".json")))
    s = g["synthetic"]
    st = mbdata.synthetic_states(s["ntaxa"], s["nsites"], nst, s["seed"], s["p_mut"], s["p_gap"])
    tr = mbtree.parse_newick(g["newick"])
    o, wall = refrun.run_mb(refrun.REF_MB_AMD, refrun.model_nexus(kind, st, tr, ngen=3000, beagle="dynamic", fixed_topology=True, precision="double"), env={"MBAMD_STATS": "1"})
    print(case, "double", "3000 generations", "%.2f s wall" % wall)
    print("\n".join(l for l in o.splitlines() if "[mbamd]" in l or "Impl Name" in l))
PY
