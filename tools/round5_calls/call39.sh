#!/bin/bash
# round 5, call 39: accumulate kernels read their source lists through LDS; the tree walk instantiated for 3, 5, 6, 7, 9, 10 states
# (standard characters) -- GPU parity, the standard-data soak, the kernels of the soak
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "other_state_counts or single_operation or scale or golden_always or general_state" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_std_dropin.py tests/test_more_datatypes_dropin.py tests/test_reports_dropin.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python - <<'PY' 2>&1 | tee gpurun_out/r5c39_soak.log
import os, sys, re
sys.path.insert(0, os.getcwd())
from tests import std_cases
from tools import refrun
kw = dict(std_cases.BIG, ngen=20000)
nex = std_cases.synthetic_nexus(beagle="dynamic", **kw).replace("nchains=1", "nchains=2").replace(" startvals tau=t V=t;\n", "")
for env in ({"MBAMD_STATS": "1"}, {"MBAMD_DEVICE_STD": "0"}):
    out, wall = refrun.run_mb(os.path.join(os.getcwd(), "oracle", "_ref", "mb_amd_full"), nex, timeout=850, env=env)
    print(env, "completed" if "Analysis completed" in out else "FAILED", "wall %.1f s" % wall)
    print("\n".join(l for l in out.splitlines() if re.match(r"\s+20000 -- ", l) or "CPU time" in l)[:700])
    seen = False
    for l in out.splitlines():
        if "[mbamd] instance 4" in l: seen = True
        if seen and "[mbamd]" in l: print(l)
PY
bash tools/round5_calls/call38.sh 2>&1 | head -16
