#!/bin/bash
# Round 4, GPU call 23: the standard-data part of the soak (all bindings, default moves), and the standard-data GPU tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r4_call23.log; : > $OUT
timeout 900 python -m pytest tests/test_std_dropin.py -x -q -m gpu 2>&1 | tail -3 | tee -a $OUT
echo "== standard data 100 taxa x 2000 characters (2 ... 6 states, gamma), 50 000 generations x 2 chains, default moves, all bindings" | tee -a $OUT
timeout 900 python - <<'PY' 2>&1 | tee -a $OUT
import os, sys, re
sys.path.insert(0, os.getcwd())
from tests import std_cases
from tools import refrun
kw = dict(std_cases.BIG, ngen=50000)
nex = std_cases.synthetic_nexus(beagle="dynamic", **kw).replace("nchains=1", "nchains=2").replace(" startvals tau=t V=t;\n", "")
for env in ({"MBAMD_STATS": "1"},):
    out, wall = refrun.run_mb(os.path.join(os.getcwd(), "oracle", "_ref", "mb_amd_full"), nex, timeout=850, env=env)
    print(env, "completed" if "Analysis completed" in out else "FAILED", "wall %.1f s" % wall)
    print("\n".join(l for l in out.splitlines() if re.match(r"\s+50000 -- ", l) or "standard data" in l or "rror" in l or "Analysis used" in l)[:900])
PY
