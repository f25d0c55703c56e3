#!/bin/bash
# Round 4, GPU call 33: the eigen-solver with one barrier per Jacobi step (two copies of A) -- tests, solver timing against MBAMD_EIGEN_256=1, codon M3 MCMC
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call33.log; : > $OUT
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_eigen_binding.py -x -q -m gpu -k "eigen" 2>&1 | tail -2 | tee -a $OUT
echo "== 1 024 threads" | tee -a $OUT; timeout 600 python tools/eigen_time.py m3 2>&1 | tail -5 | tee -a $OUT
echo "== 256 threads (MBAMD_EIGEN_256=1)" | tee -a $OUT; MBAMD_EIGEN_256=1 timeout 600 python tools/eigen_time.py m3 2>&1 | tail -5 | tee -a $OUT
timeout 900 python - <<'PY' 2>&1 | tee -a $OUT
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
from mrbayes_amd import data as mbdata, tree as mbtree
from tools import refrun
g5 = json.load(open(os.path.join(bench.GOLD, "bench_c5.json")))
s5 = g5["synthetic"]
st5 = mbdata.synthetic_states(s5["ntaxa"], s5["nsites"], 61, s5["seed"], s5["p_mut"], s5["p_gap"])
tr5 = mbtree.parse_newick(g5["newick"])
for env in ({}, {"MBAMD_EIGEN_256": "1"}, {}):
    walls = []
    for ngen in (500, 4500):
        o, wall = refrun.run_mb(refrun.REF_MB_AMD_FULL, refrun.model_nexus("m3", st5, tr5, ngen=ngen, beagle="dynamic", fixed_topology=True), env=env)
        assert "Analysis completed" in o
        walls.append(wall)
    print(env, "codon M3 fixed topology, all bindings: %.0f generations/s" % (4000 / (walls[1] - walls[0])))
PY
