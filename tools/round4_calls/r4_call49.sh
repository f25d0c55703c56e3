#!/bin/bash
# Round 4, GPU call 49: whole-MCMC generations/s of the unmodified MrBayes binary with beagleprecision=double (codon M3 100 x 5 000 and protein 200 x 10 000, topology fixed)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call49.log; : > $OUT
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
    for env in ({}, {"MBAMD_F64_MFMA_NO_LDS": "1", "MBAMD_F64_NO_TIPS_KERNEL": "1", "MBAMD_F64_NO_MATRIX_QUEUE": "1", "MBAMD_F64_NO_RING": "1"}):
        walls = []
        for ngen in (500, 4500):
            o, wall = refrun.run_mb(refrun.REF_MB_AMD, refrun.model_nexus(kind, st, tr, ngen=ngen, beagle="dynamic", fixed_topology=True, precision="double"), env=env)
            assert "Analysis completed" in o, o[-2000:]
            walls.append(wall)
        print(case, "beagleprecision=double, topology fixed,", "this round's kernels and staging" if not env else "the plain level kernels, a synchronisation per staged list (round 4 mid-way)", ": %.0f generations/s" % (4000 / (walls[1] - walls[0])), "(double-precision engine)" if "double-precision" in o else "(NOT the double-precision engine)")
PY
