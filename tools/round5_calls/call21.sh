#!/bin/bash
# round 5, call 21: where an evaluation of the standard-data soak spends its host time (MBAMD_STATS=1), 5 000 generations x 2 chains
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/r5c21_std_stats.log
import os, sys, re
sys.path.insert(0, os.getcwd())
from tests import std_cases
from tools import refrun
kw = dict(std_cases.BIG, ngen=5000)
nex = std_cases.synthetic_nexus(beagle="dynamic", **kw).replace("nchains=1", "nchains=2").replace(" startvals tau=t V=t;\n", "")
out, wall = refrun.run_mb(os.path.join(os.getcwd(), "oracle", "_ref", "mb_amd_full"), nex, timeout=550, env={"MBAMD_STATS": "1"})
print("wall %.1f s" % wall)
print("\n".join(l for l in out.splitlines() if "[mbamd]" in l or "CPU time" in l)[:6000])
PY
